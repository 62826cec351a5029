#!/usr/bin/env python3
"""Host enqueue time of ONE training step against its GPU time, at the per-rank batch (is the small-patch regime host-bound?).
The queue is drained before each timed enqueue, so nothing blocks on queue depth.   python tools/host_enqueue.py B=16 S=25,35,45,64"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drs_amd.net import DilatedNet
from drs_amd import patches as P
from drs_amd.synthetic import make_tile, grid_instances


def main(B, sizes):
    dev = "cuda:0"
    tile, lab = make_tile(1024, 1024, 5, 6, seed=1234)
    pool = P.TilePool([tile], [lab], dev)
    net = DilatedNet("dilated_grsl_rate8", 5, 6, 0.005, b_max=B, s_max=max(sizes), device=dev)
    np.random.seed(0)
    for S in sizes:
        inst = grid_instances(1024, 1024, S, 25, 4096, seed=0)

        def prep(i):
            rows = inst[(i * B) % 4000:(i * B) % 4000 + B]
            aug = P.draw_augmentation(rows, S, 5, noise="device")
            P.crop_to_net(net, pool, rows, S, [0.5] * 3, [0.2] * 3, aug)

        for i in range(5):
            prep(i); net.train_step(B, S, 0.01)
        torch.cuda.synchronize()
        hp, hs, gp = [], [], []
        for i in range(20):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            prep(i)
            t1 = time.perf_counter()
            e0.record()
            net.train_step(B, S, 0.01)
            e1.record()
            t2 = time.perf_counter()
            torch.cuda.synchronize()
            hp.append(t1 - t0); hs.append(t2 - t1); gp.append(e0.elapsed_time(e1))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(40):
            prep(i); net.train_step(B, S, 0.01)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40
        print("B=%d S=%d: host prep (draw + crop enqueue) %.3f ms, host step enqueue %.3f ms, GPU step (events) %.3f ms, pipelined step %.3f ms"
              % (B, S, 1e3 * np.median(hp), 1e3 * np.median(hs), np.median(gp), 1e3 * dt), flush=True)


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    main(int(kw.get("B", 16)), [int(v) for v in kw.get("S", "25,35,45,64").split(",")])
