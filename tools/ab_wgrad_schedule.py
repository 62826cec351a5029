#!/usr/bin/env python3
"""VERDICT r05 item 4 -- ONE structural experiment on the per-rank regime: the filter gradients of a step deferred / grouped.

In-process A/B of one rank's training step (development library, no collectives), arms interleaved on one device:
    p        the product's schedule: two gz slabs in turn, ONE filter-gradient stream, block i+1's filter gradient issued beside block i's
             batch-norm backward / input gradient; the chain waits for block i+2's filter gradient before it rewrites a gz slab
    L<n>e    a gz slab and a split slab PER LAYER (the chain never waits for a filter gradient), n filter-gradient streams (layer i on stream
             i % n: the launches overlap each other's fill and drain, which is what one grouped launch over the layers would do), issued as
             the chain goes
    L<n>d    the same, but every filter gradient waits for the END of the chain (the chain runs uncontended, then the "group")
    L<n>o    issued after the chain's last launch, each behind its own gz only (host order of the deferred form, dependencies of the eager one)
    L<n>a    issued as the chain goes, but block i's filter gradient is released by the END of block i's input gradient: it runs beside block
             i-1's elementwise passes (which leave the matrix pipe idle) and input gradient.  With two=1 the L-arms force the two-stream pass
             at any batch (at B = 128 the product runs ONE stream: `p` is then that)
Every arm runs the same kernels on the same operands: the gradients after a step are compared bit for bit with the product's.

    python tools/ab_wgrad_schedule.py [B=16] [S=25,35,45,55,64,65,75,85] [arms=p,L1e,L2e,L4e,L1d,L2d,L4d,L4o] [steps=20] [rounds=4]
"""
import os
import re
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drs_amd import _lib  # noqa: E402
d = _lib.dev()
_lib._lib = d.lib          # the whole net on libdrs_hip_dev.so
from drs_amd.net import DilatedNet  # noqa: E402
from drs_amd import patches as P  # noqa: E402
from drs_amd.synthetic import make_tile, grid_instances  # noqa: E402


def parse(arm):
    if arm == "p":
        return 0, 1, 0
    m = re.fullmatch(r"L(\d)([edoa])", arm)
    return 1, int(m.group(1)), {"e": 0, "d": 1, "o": 2, "a": 3}[m.group(2)]


def main(B=16, Ss=(64,), arms=("p", "L1e"), steps=20, rounds=4, two=None):
    dev = "cuda:0"
    tile, lab = make_tile(1024, 1024, 5, 6, seed=1234)      # (instances wrap around: 4096 of them serve any batch)
    pool = P.TilePool([tile], [lab], dev)
    smax = max(Ss)
    nets = {}
    for slabs in sorted({parse(a)[0] for a in arms}):
        d.drs_debug_wgrad_schedule(slabs, -1, -1)
        nets[slabs] = DilatedNet("dilated_grsl_rate8", 5, 6, 0.005, b_max=B, s_max=smax, device=dev, seed=42)
        if slabs and two is not None:
            nets[slabs].set_two_streams(two)
    d.drs_debug_wgrad_schedule(0, -1, -1)
    weighted = {a: 0.0 for a in arms}
    for S in Ss:
        inst = grid_instances(1024, 1024, S, 25, 4096, seed=0)

        def step(net, i, update=True):
            rows = inst[(i * B) % 4000:(i * B) % 4000 + B]
            aug = P.draw_augmentation(rows, S, 5, noise="device")
            P.crop_to_net(net, pool, rows, S, [0.5] * 3, [0.2] * 3, aug)
            return net.train_step(B, S, 0.01, apply_update=update)
        # bitwise: one step without update from the same variables, every arm against the product's gradients
        ref = None
        for a in arms:
            slabs, ns, defer = parse(a)
            net = nets[slabs]
            net.params.copy_(nets[min(nets)].params)
            net.bn.copy_(nets[min(nets)].bn)
            d.drs_debug_wgrad_schedule(-1, ns, defer)
            np.random.seed(1)
            out = step(net, 0, update=False)
            torch.cuda.synchronize()
            got = (net.grads.clone(), float(net.loss_value(out["loss_parts"])))
            if ref is None:
                ref = got
            assert torch.equal(got[0], ref[0]) and got[1] == ref[1], "arm %s differs from the first arm at S = %d" % (a, S)
        best = {a: [] for a in arms}
        for r in range(rounds + 1):
            for a in arms:
                slabs, ns, defer = parse(a)
                net = nets[slabs]
                d.drs_debug_wgrad_schedule(-1, ns, defer)
                np.random.seed(0)
                for i in range(3):
                    step(net, i)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(steps):
                    step(net, i)
                torch.cuda.synchronize()
                if r:
                    best[a].append((time.perf_counter() - t0) / steps * 1e3)
        for a in arms:
            weighted[a] += float(np.median(best[a]))
        print("B=%d S=%d  " % (B, S) + "   ".join("%s %.3f (min %.3f)" % (a, float(np.median(v)), min(v)) for a, v in best.items()) + "   [gradients bitwise equal]",
              flush=True)
    print("size-weighted patches/s per rank (sizes drawn uniformly over the listed ones): " +
          "   ".join("%s %.0f" % (a, B * len(Ss) / (weighted[a] * 1e-3)) for a in arms), flush=True)
    d.drs_debug_wgrad_schedule(0, 1, 0)


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    main(int(kw.get("B", 16)), tuple(int(v) for v in kw.get("S", "64").split(",")), tuple(kw.get("arms", "p,L1e,L2e,L4e,L1d,L2d,L4d,L4o").split(",")),
         int(kw.get("steps", 20)), int(kw.get("rounds", 4)), int(kw["two"]) if "two" in kw else None)
