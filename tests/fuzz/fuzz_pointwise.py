#!/usr/bin/env python3
"""Randomised shapes through the BN / activation / pool forward and backward entry points and the classifier + loss block, against
the fp64 oracle: the checks are those of tests/test_gpu_ops.py (called as functions), the shapes are drawn here -- channels 4..576 in
steps of 4, sides 1..40, batches 1..6, halos 0..8, pool on / off, ReLU / leaky.  Test infrastructure.
    python tests/fuzz/fuzz_pointwise.py [n=300] [seed=0]"""
import os, sys, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from drs_amd import _lib
import test_gpu_ops as G


def main(n=300, seed=0):
    rng = np.random.default_rng(seed)
    nbad = 0
    for i in range(n):
        if i % 4 != 3:
            C = int(rng.choice([4, 8, 16, 32, 48, 64, 96, 100, 128, 160, 192, 256, 320, 448, 512, 576]))
            S = int(rng.integers(1, 41))
            B = int(rng.integers(1, 7))
            if B * S * S < 16:            # (the test feeds its statistics through an fp32 (sum, sum of squares) slab: meaningless on a handful of pixels)
                S = 4
            while B * S * S * C > 2.5e6:
                S = max(1, S - 3)
            args = (C, int(rng.integers(0, 2)), float(rng.choice([0.0, 0.1])), B, S, int(rng.integers(0, 9)))
            name, fn = "bn_act_pool", G.test_bn_act_pool_forward_backward
        else:
            C = int(rng.choice([64, 128, 192, 256, 448]))
            K = int(rng.integers(2, 9))
            S = int(rng.integers(2, 30))
            B = int(rng.integers(1, 6))
            args = (C, K, B, S, int(rng.choice([0, 0, 2, 6])), bool(rng.integers(0, 2)))
            name, fn = "classifier", G.test_classifier_loss
        try:
            fn(_lib, *args)
            if i % 25 == 0:
                print("ok  ", name, args, flush=True)
        except ZeroDivisionError:       # (a random mask that leaves no pixel: the test's own 1 / n)
            continue
        except AssertionError:
            nbad += 1
            tb = traceback.format_exc().strip().splitlines()
            print("FAIL", name, args, "|", tb[-3].strip()[:150], "|", tb[-1][:200], flush=True)
    print("%d cases, %d failed" % (n, nbad))
    sys.exit(1 if nbad else 0)


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    main(int(kw.get("n", 300)), int(kw.get("seed", 0)))
