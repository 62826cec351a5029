"""-m gpu: a short, seeded run of every randomised-shape checker under tests/fuzz/ (the long runs are in profiles/r03/fuzz.txt):
convolution entry points, BN / pool / classifier entry points, crop + stitch kernels, whole nets -- all against the oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("script,args", [("fuzz_ops.py", ["n=40", "seed=101"]), ("fuzz_pointwise.py", ["n=80", "seed=102"]),
                                          ("fuzz_patches.py", ["n=60", "seed=103"]), ("fuzz_nets.py", ["n=12", "seed=104"])])
def test_randomised_shapes_against_the_oracle(script, args):
    r = subprocess.run([sys.executable, os.path.join(HERE, "fuzz", script)] + args, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert "0 failed" in r.stdout
