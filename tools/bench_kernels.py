#!/usr/bin/env python3
"""Per-layer timing of the conv kernels (forward, dgrad, wgrad) on the Dilated8Pooling shapes: development aid for
kernel tuning (HIP events on the launch stream, interleaved repetitions, median)."""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drs_amd import _lib  # noqa: E402
_lib = _lib.dev()      # libdrs_hip_dev.so: the library with the A/B switches of include/drs_dev.h
from drs_amd.nets import Plan  # noqa: E402

DEV = "cuda:0"


def timeit(fn, reps=7):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def main(B=128, S=64, net="dilated_grsl_rate8", which="fwd,dgrad,wgrad", pad0=32):
    _lib.load()
    plan = Plan(net, 5, 6, first_cin_pad=pad0)
    st = torch.cuda.current_stream(DEV).cuda_stream
    M = B * S * S
    tot = {}
    for i, L in enumerate(plan.layers):
        P = L.halo
        x = torch.randn(B * (S + 2 * P) ** 2 * L.cin_k, device=DEV)
        g = torch.randn(B * (S + 2 * P) ** 2 * L.cout, device=DEV)
        w = torch.randn(-(-L.k * L.k * L.cin_k // 32) * 32 * L.cout, device=DEV) * 0.05      # rows padded to whole K-steps (conv1 on 8 channels)
        wt = torch.randn(L.k * L.k * L.cin_k * L.cout, device=DEV) * 0.05
        bias = torch.zeros(L.cout, device=DEV)
        z = torch.zeros(M * max(L.cout, L.cin_k), device=DEV)
        mt = _lib.query("drs_conv_mtile", L.cout)
        stats = torch.zeros(((M + mt - 1) // mt) * L.cout * 2, device=DEV)
        ns = _lib.query("drs_conv_wgrad_splits", B, S, L.k, L.cin_k, L.cout)
        slab = torch.zeros(ns * L.k * L.k * L.cin_k * L.cout, device=DEV)
        gw = torch.zeros(L.k * L.k * L.cin_k * L.cout, device=DEV)
        fl = 2.0 * M * L.k * L.k * L.cin_k * L.cout
        row = "%-6s k%d r%d %3d->%3d " % (L.name, L.k, L.rate, L.cin_k, L.cout)
        if "ab" in which:       # A/B of the halo-tap skipping inside one process (same device, interleaved)
            f = lambda: _lib.call("drs_conv_forward", x.data_ptr(), B, S, P, L.cin_k, 0, w.data_ptr(), bias.data_ptr(), L.k, L.rate,
                                  L.pad_b, L.cin_k, L.cout, z.data_ptr(), L.cout, 0, 0, stats.data_ptr(), st)
            res = {0: [], 1: []}
            for rep in range(4):
                for v in (0, 1):
                    _lib.load().drs_debug_skip_taps(2 * v)
                    res[v].append(timeit(f, reps=3))
            _lib.load().drs_debug_skip_taps(1)
            row += " fwd all taps %6.3f ms | skipping halo tap rows %6.3f ms (%+.1f %%)" % (min(res[0]), min(res[1]), 100 * (min(res[1]) / min(res[0]) - 1))
        if "wsp" in which:      # filter-gradient time against the number of workgroups the pixel split aims at
            for target in (768, 1152, 1536, 2304):
                _lib.load().drs_debug_wgrad_target(target)
                ns2 = _lib.query("drs_conv_wgrad_splits", B, S, L.k, L.cin_k, L.cout)
                slab2 = torch.zeros(ns2 * L.k * L.k * L.cin_k * L.cout, device=DEV)
                ms = timeit(lambda: _lib.call("drs_conv_wgrad", x.data_ptr(), B, S, P, L.cin_k, 0, g.data_ptr(), P, L.cout, 0, L.k, L.rate, L.pad_b,
                                              L.cin_k, L.cin_k, L.cout, slab2.data_ptr(), gw.data_ptr(), st), reps=5)
                row += " | %d: %d splits %6.3f ms" % (target, ns2, ms)
                del slab2
            _lib.load().drs_debug_wgrad_target(2048)
        if "wab" in which:      # A/B of skipping the all-halo pixel chunks in the filter gradient, inside one process
            f = lambda: _lib.call("drs_conv_wgrad", x.data_ptr(), B, S, P, L.cin_k, 0, g.data_ptr(), P, L.cout, 0, L.k, L.rate, L.pad_b,
                                  L.cin_k, L.cin_k, L.cout, slab.data_ptr(), gw.data_ptr(), st)
            res = {0: [], 1: []}
            for rep in range(4):
                for v in (0, 1):
                    _lib.load().drs_debug_skip_taps(2 * v)
                    res[v].append(timeit(f, reps=3))
            _lib.load().drs_debug_skip_taps(1)
            row += " wgrad all chunks %6.3f ms | skipping all-halo chunks %6.3f ms (%+.1f %%)" % (min(res[0]), min(res[1]), 100 * (min(res[1]) / min(res[0]) - 1))
        if "fwd" in which:
            ms = timeit(lambda: _lib.call("drs_conv_forward", x.data_ptr(), B, S, P, L.cin_k, 0, w.data_ptr(), bias.data_ptr(), L.k, L.rate,
                                          L.pad_b, L.cin_k, L.cout, z.data_ptr(), L.cout, 0, 0, stats.data_ptr(), st))
            row += " fwd %6.3f ms %6.1f TF" % (ms, fl / ms / 1e9)
            tot["fwd"] = tot.get("fwd", 0) + ms
        if "dgrad" in which and i > 0:
            ms = timeit(lambda: _lib.call("drs_conv_forward", g.data_ptr(), B, S, P, L.cout, 0, wt.data_ptr(), None, L.k, L.rate, L.pad_a,
                                          L.cout, L.cin_k, z.data_ptr(), L.cin_k, 0, 0, None, st))
            row += " dgrad %6.3f ms %6.1f TF" % (ms, fl / ms / 1e9)
            tot["dgrad"] = tot.get("dgrad", 0) + ms
        if "wgrad" in which:
            ms = timeit(lambda: _lib.call("drs_conv_wgrad", x.data_ptr(), B, S, P, L.cin_k, 0, g.data_ptr(), P, L.cout, 0, L.k, L.rate, L.pad_b,
                                          L.cin_k, L.cin_k, L.cout, slab.data_ptr(), gw.data_ptr(), st))
            row += " wgrad %6.3f ms %6.1f TF (splits %d)" % (ms, fl / ms / 1e9, ns)
            tot["wgrad"] = tot.get("wgrad", 0) + ms
        print(row, flush=True)
    print("total ms:", {k: round(v, 2) for k, v in tot.items()})


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    main(int(kw.get("B", 128)), int(kw.get("S", 64)), kw.get("net", "dilated_grsl_rate8"), kw.get("which", "fwd,dgrad,wgrad"),
         int(kw.get("pad0", 32)))
