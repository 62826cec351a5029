"""CPU, build container only: a short, seeded run of tests/fuzz/fuzz_vs_reference.py -- the oracle's host half (oracle/host_ref.py) and
the product's host functions against the reference's OWN functions, imported from /root/reference the way
tests/golden/make_goldens.py imports them, on random inputs under the same seeds, exact equality.  Skipped where the reference is not
present (the GPU box: nothing of the reference travels).  The long runs are recorded in profiles/r03/fuzz.txt."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference exists only in the build container")
def test_host_functions_against_the_imported_reference():
    r = subprocess.run([sys.executable, os.path.join(HERE, "fuzz", "fuzz_vs_reference.py"), "n=40", "seed=77"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert "40 rounds against the reference's own functions: 0 mismatches" in r.stdout
