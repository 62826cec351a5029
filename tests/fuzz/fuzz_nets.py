#!/usr/bin/env python3
"""Randomised (net type, bands, classes, batch, side) through the whole-net checks of tests/test_gpu_net.py (eval and train parity
against the fp64 oracle: logits, loss, decision margins, every gradient, moving statistics, confusion matrix) and
tests/test_gpu_engine.py (the step engine bitwise equal to the op-level sequence), called as functions.  Test infrastructure.
    python tests/fuzz/fuzz_nets.py [n=60] [seed=0]"""
import os, sys, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_net as N
import test_gpu_engine as E
from drs_amd.nets import known_net_types


def main(n=60, seed=0):
    rng = np.random.default_rng(seed)
    nets = known_net_types()
    nbad = 0
    for i in range(n):
        net = str(rng.choice(nets))
        ch = int(rng.choice([3, 4, 5]))
        K = int(rng.choice([2, 6, 7]))
        B = int(rng.integers(1, 5))
        S = int(rng.integers(5, 27))
        args = (net, ch, K, B, S)
        for name, fn in (("parity", N.test_eval_and_train_parity), ("engine==op-level", E.test_engine_equals_op_level_path_bitwise)):
            try:
                fn(*args)
            except AssertionError:
                nbad += 1
                tb = traceback.format_exc().strip().splitlines()
                print("FAIL", name, args, "|", " | ".join(t.strip()[:140] for t in tb[-4:]), flush=True)
        if i % 5 == 0:
            print("done", i, args, flush=True)
    print("%d nets x 2 checks, %d failed" % (n, nbad))
    sys.exit(1 if nbad else 0)


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    main(int(kw.get("n", 60)), int(kw.get("seed", 0)))
