#!/usr/bin/env python3
"""Same command line as the reference's contest_dilated_random.py; the work runs on the MI355X path (drs_amd.cli)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

if __name__ == "__main__":
    from drs_amd.cli import main_contest
    main_contest(sys.argv)
