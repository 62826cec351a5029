#!/usr/bin/env python3
"""In-process A/B at LARGE batches (multi-round launches): the backward pass on one stream (the product's rule from 2^18 pixels) against
two streams with the filter-gradient stream at the highest (the product's choice for small steps), the caller's or the LOWEST stream
priority.  At 128 x 64 x 64 the backward elementwise passes (bn_bwd_reduce + bn_bwd_apply: 3.2 ms of the 47.7 ms step) run with the
matrix pipe idle; a filter gradient of many rounds running BESIDE them could hide them -- if the dispatcher hands freed workgroup
slots to the chain first, i.e. with the filter-gradient stream at the LOWER priority (with it at the higher one the chain's kernels
only get slots when the filter gradient has none left to dispatch: a serial schedule with extra barriers).
    python tools/ab_two_streams_big.py [B=128] [S=64] [steps=20] [rounds=4] [arms=one,hi,same,lo]
Development library (drs_debug_wg_stream_prio is read when a net makes its filter-gradient stream)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drs_amd import _lib  # noqa: E402
d = _lib.dev()
_lib._lib = d.lib
from drs_amd.net import DilatedNet  # noqa: E402
from drs_amd import patches as P  # noqa: E402
from drs_amd.synthetic import make_tile, grid_instances  # noqa: E402

ARMS = {"one": (0, 2), "hi": (1, 2), "same": (1, 0), "lo": (1, 1)}      # (two streams, drs_debug_wg_stream_prio arm)


def main(B=128, S=64, steps=20, rounds=4, arms=("one", "hi", "same", "lo"), chain=-1):
    dev = "cuda:0"
    tile, lab = make_tile(2048, 2048, 5, 6, seed=1234)
    pool = P.TilePool([tile], [lab], dev)
    inst = grid_instances(2048, 2048, S, 25, 8192, seed=0)
    nets = {}
    for a in arms:
        two, prio = ARMS[a]
        d.drs_debug_wg_stream_prio(prio)
        nets[a] = DilatedNet("dilated_grsl_rate8", 5, 6, 0.005, b_max=B, s_max=S, device=dev, seed=42)
        nets[a].set_two_streams(two)

        def step(i, net=nets[a]):
            rows = inst[(i * B) % 8000:(i * B) % 8000 + B]
            aug = P.draw_augmentation(rows, S, 5, noise="device")
            P.crop_to_net(net, pool, rows, S, [0.5] * 3, [0.2] * 3, aug)
            return net.train_step(B, S, 0.01)
        np.random.seed(0)
        step(0)                  # the stream is made here, at the priority set above
        torch.cuda.synchronize()
    d.drs_debug_wg_stream_prio(2)
    d.drs_debug_chain_mode(chain)
    ref = None
    for a in arms:               # the same kernels on the same operands: one more step from equal variables must give equal bits
        nets[a].params.copy_(nets[arms[0]].params); nets[a].mom.copy_(nets[arms[0]].mom); nets[a].bn.copy_(nets[arms[0]].bn)
    best = {a: [] for a in arms}
    for r in range(rounds + 1):
        for a in arms:
            net = nets[a]

            def step(i):
                rows = inst[(i * B) % 8000:(i * B) % 8000 + B]
                aug = P.draw_augmentation(rows, S, 5, noise="device")
                P.crop_to_net(net, pool, rows, S, [0.5] * 3, [0.2] * 3, aug)
                return net.train_step(B, S, 0.01)
            np.random.seed(r)
            for i in range(2):
                step(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                step(i)
            torch.cuda.synchronize()
            if r:
                best[a].append((time.perf_counter() - t0) / steps * 1e3)
        got = [nets[a].params.clone() for a in arms]
        assert all(torch.equal(g, got[0]) for g in got), "arms differ after round %d" % r
    print("B=%d S=%d chain_mode=%d  " % (B, S, chain) + "   ".join("%s %.3f ms (min %.3f) = %.0f patches/s" % (a, float(np.median(v)), min(v), B / (np.median(v) * 1e-3))
                                                                  for a, v in best.items()) + "   [variables bitwise equal]", flush=True)
    d.drs_debug_chain_mode(-1)


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    main(int(kw.get("B", 128)), int(kw.get("S", 64)), int(kw.get("steps", 20)), int(kw.get("rounds", 4)), tuple(kw.get("arms", "one,hi,same,lo").split(",")),
         int(kw.get("chain", -1)))
