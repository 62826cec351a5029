#!/usr/bin/env python3
"""The population comparison of tests/test_gpu_accuracy_parity.py over MANY seeds (development aid: is a mean paired difference of
the late training loss / held-out accuracy between the HIP path and the CPU oracle chance or bias?).
    python tests/fuzz/parity_population.py [seeds=24] [first=100]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import conftest
torch.set_num_threads(conftest._granted_cores())
import test_gpu_accuracy_parity as T
from drs_amd.synthetic import make_tile

kw = dict(a.split("=") for a in sys.argv[1:])
n, first = int(kw.get("seeds", 24)), int(kw.get("first", 100))
tile, lab = make_tile(160, 160, T.CH, T.K, seed=3, n_seeds=24, class_signal=0.6)
held, held_lab = make_tile(96, 96, T.CH, T.K, seed=4, n_seeds=12, class_signal=0.6)
mean, std = tile[:, :, :3].mean(axis=(0, 1)), tile[:, :, :3].std(axis=(0, 1))
rows = []
for seed in range(first, first + n):
    ld, lt, ad, at = T._run(seed, tile, lab, held, held_lab, mean, std)
    rows.append((ad - at, np.mean(ld[-20:]) - np.mean(lt[-20:]), np.mean(ld[-20:]), np.mean(lt[-20:]), np.mean(ld[20:60]) - np.mean(lt[20:60])))
    print("seed %d  accuracy HIP %.4f oracle %.4f   late loss HIP %.4f oracle %.4f" % (seed, ad, at, rows[-1][2], rows[-1][3]), flush=True)
r = np.asarray(rows)
for name, col in (("held-out accuracy", 0), ("late loss (steps 100-119)", 1), ("loss, steps 20-59", 4)):
    d = r[:, col]
    print("%-28s paired difference HIP - oracle: mean %+.4f, standard error %.4f (%.1f sigma), %d of %d positive" % (
        name, d.mean(), d.std(ddof=1) / np.sqrt(len(d)), d.mean() / (d.std(ddof=1) / np.sqrt(len(d))), int((d > 0).sum()), len(d)))
