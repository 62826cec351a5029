#!/usr/bin/env python3
"""Host side of one training step at the per-rank batch of 8 GPUs (16 patches): library calls per step and host time per step with
an empty queue, for the step-level path (DilatedNet -> drs_train_step) and the op-level path (engine=False).
    python tools/host_profile.py [B=16]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drs_amd import _lib, patches as P
from drs_amd.net import DilatedNet
from drs_amd.synthetic import make_tile, grid_instances

def main(B=16, S=64):
    dev = "cuda:0"
    tile, lab = make_tile(1024, 1024, 5, 6, seed=1234)
    pool = P.TilePool([tile], [lab], dev)
    inst = grid_instances(1024, 1024, S, 25, 4096, seed=0)
    calls = [0]
    real_call = _lib.call
    def counting(name, *a):
        calls[0] += 1
        return real_call(name, *a)
    for engine in (True, False):
        net = DilatedNet("dilated_grsl_rate8", 5, 6, 0.005, b_max=B, s_max=S, device=dev, engine=engine)
        np.random.seed(0)
        def step(i):
            rows = inst[(i * B) % 4000:(i * B) % 4000 + B]
            aug = P.draw_augmentation(rows, S, 5, noise="device")
            P.crop_to_net(net, pool, rows, S, [0.5] * 3, [0.2] * 3, aug)
            return net.train_step(B, S, 0.01)
        for i in range(5):
            step(i)
        _lib.call = counting
        P._lib.call = counting
        import drs_amd.net as N, drs_amd.engine as E
        calls[0] = 0
        ts = []
        for i in range(10):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            step(i)
            ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        _lib.call = real_call
        P._lib.call = real_call
        print("%-28s library calls per step: %5.1f   host time per step with an empty queue: median %.2f ms"
              % ("step-level (drs_train_step)" if engine else "op-level (engine=False)", calls[0] / 10.0, 1e3 * np.median(ts)))
        del net

if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    main(int(kw.get("B", 16)))
