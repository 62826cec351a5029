#!/usr/bin/env python3
"""Time the exact-fp32 filter-gradient kernel on the Dilated8Pooling shapes (development aid; use under rocprofv3 --pmc)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drs_amd import _lib
_lib = _lib.dev()      # libdrs_hip_dev.so: the library with the A/B switches of include/drs_dev.h
from drs_amd.nets import Plan
DEV = "cuda:0"
def main(B=128, S=64, layers="4,8", reps=3, skip=1, target=0, fwd=0, variant=-1):
    _lib.load()
    _lib.load().drs_debug_skip_taps(skip)
    if target:
        _lib.load().drs_debug_wgrad_target(target)
    plan = Plan("dilated_grsl_rate8", 5, 6, first_cin_pad=32)
    st = torch.cuda.current_stream(DEV).cuda_stream
    for i, L in enumerate(plan.layers):
        if str(i + 1) not in layers.split(","):
            continue
        P = L.halo
        x = torch.randn(B * (S + 2 * P) ** 2 * L.cin_k, device=DEV)
        g = torch.randn(B * (S + 2 * P) ** 2 * L.cout, device=DEV)
        ns = _lib.query("drs_conv_wgrad_splits", B, S, L.k, L.cin_k, L.cout)
        slab = torch.zeros(ns * L.k * L.k * L.cin_k * L.cout, device=DEV)
        gw = torch.zeros(L.k * L.k * L.cin_k * L.cout, device=DEV)
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.call("drs_conv_wgrad", x.data_ptr(), B, S, P, L.cin_k, 0, g.data_ptr(), P, L.cout, 0, L.k, L.rate, L.pad_b,
                      L.cin_k, L.cin_k, L.cout, slab.data_ptr(), gw.data_ptr(), st)
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        fl = 2.0 * B * S * S * L.k * L.k * L.cin_k * L.cout
        print("%s wgrad f32 %.3f ms %.1f TF (splits %d)" % (L.name, min(ts), fl / min(ts) / 1e9, ns), flush=True)
        if fwd:
            w = torch.randn(L.k * L.k * L.cin_k * L.cout, device=DEV) * 0.05
            z = torch.zeros(B * S * S * L.cout, device=DEV)
            for _ in range(reps):
                _lib.call("drs_conv_forward", x.data_ptr(), B, S, P, L.cin_k, 0, w.data_ptr(), None, L.k, L.rate, L.pad_b, L.cin_k, L.cout,
                          z.data_ptr(), L.cout, 0, 0, None, st)
            torch.cuda.synchronize()
if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    main(int(kw.get("B", 128)), int(kw.get("S", 64)), kw.get("layers", "4,8"), skip=int(kw.get("skip", 1)), target=int(kw.get("target", 0)),
         fwd=int(kw.get("fwd", 0)), variant=int(kw.get("variant", -1)), reps=int(kw.get("reps", 3)))
