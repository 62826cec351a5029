#!/usr/bin/env python3
"""Randomised shapes through the opt-in split-bf16 convolution entry points (2 / 3 terms; forward, input gradient, filter gradient)
with the checks of tests/test_gpu_split.py, called as a function.  Test infrastructure.
    python tests/fuzz/fuzz_split.py [n=80] [seed=0]"""
import os, sys, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from drs_amd import _lib
import test_gpu_split as G


def main(n=80, seed=0):
    rng = np.random.default_rng(seed)
    nbad = 0
    for i in range(n):
        k = int(rng.integers(1, 6))
        rate = int(rng.integers(1, 9)) if k > 1 else 1
        cin = int(rng.choice([32, 64, 128, 192, 256]))
        cout = int(rng.choice([64, 128, 192, 256]))            # (the split kernels take Cout in steps of 64 and say so)
        B, S = int(rng.integers(1, 5)), int(rng.integers(4, 30))
        while B * S * S * k * k * cin * cout > 2e10:
            S = max(4, S - 4)
        ns = int(rng.choice([2, 3]))
        args = (k, rate, cin, cout, B, S, ns)
        try:
            G.test_conv_split_forward_dgrad_wgrad(_lib, *args)
            if i % 10 == 0:
                print("ok  ", args, flush=True)
        except AssertionError:
            nbad += 1
            tb = traceback.format_exc().strip().splitlines()
            print("FAIL", args, "|", " | ".join(t.strip()[:150] for t in tb[-3:]), flush=True)
    print("%d cases, %d failed" % (n, nbad))
    sys.exit(1 if nbad else 0)


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    main(int(kw.get("n", 80)), int(kw.get("seed", 0)))
