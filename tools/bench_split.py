#!/usr/bin/env python3
"""Per-layer accuracy and timing of the split-bf16 conv kernels against the exact-fp32 MFMA kernels
(development aid; Dilated8Pooling shapes, HIP events, median of interleaved repetitions)."""
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


def main(B=128, S=64, net="dilated_grsl_rate8", layers=None):
    _lib.load()
    # the slabs below are random INCLUDING their halo (only timing and the arithmetic are of interest here), so the halo-tap
    # skipping, whose decisions depend on the tile height, must be off for the error column to compare like with like
    _lib.load().drs_debug_skip_taps(0)
    plan = Plan(net, 5, 6, first_cin_pad=32)
    st = torch.cuda.current_stream(DEV).cuda_stream
    M = B * S * S
    tot = {}
    for i, L in enumerate(plan.layers):
        if layers and str(i + 1) not in layers.split(","):
            continue
        P = L.halo
        n = B * (S + 2 * P) ** 2 * L.cin_k
        zero = os.environ.get("SPLIT_ZERO") == "1"      # all-zero operands: what the kernels do when the chip need not hold its clock down
        x = torch.zeros(n, device=DEV) if zero else torch.randn(n, device=DEV)
        w = torch.zeros(L.k * L.k * L.cin_k * L.cout, device=DEV) if zero else torch.randn(L.k * L.k * L.cin_k * L.cout, device=DEV) * 0.05
        bias = torch.zeros(L.cout, device=DEV)
        z = torch.zeros(M * L.cout, device=DEV)
        fl = 2.0 * M * L.k * L.k * L.cin_k * L.cout
        row = "%-6s k%d r%d %3d->%3d " % (L.name, L.k, L.rate, L.cin_k, L.cout)
        ms = timeit(lambda: _lib.call("drs_conv_forward", x.data_ptr(), B, S, P, L.cin_k, 0, w.data_ptr(), bias.data_ptr(), L.k, L.rate,
                                      L.pad_b, L.cin_k, L.cout, z.data_ptr(), L.cout, 0, 0, None, st))
        row += " f32 %6.3f ms %6.1f TF |" % (ms, fl / ms / 1e9)
        tot["f32"] = tot.get("f32", 0) + ms
        zref = z.clone()
        # fp64 reference on a sample of output rows would need the oracle; compare against the exact-fp32 kernel instead
        for ns, variant in [(int(a[0]), int(a[1])) for a in os.environ.get('SPLIT_CASES', '20,21').split(',')]:
            _lib.load().drs_debug_variant(variant)
            xp = torch.zeros(ns * n, dtype=torch.int16, device=DEV)
            wf = torch.zeros(ns * w.numel(), dtype=torch.int16, device=DEV)
            ms_s = timeit(lambda: _lib.call("drs_split_terms", x.data_ptr(), n, ns, xp.data_ptr(), st))
            _lib.call("drs_filter_split", w.data_ptr(), L.k, L.cin_k, L.cin_k, L.cout, ns, wf.data_ptr(), None, st)
            z2 = torch.zeros(M * L.cout, device=DEV)
            ms = timeit(lambda: _lib.call("drs_conv_forward_split", xp.data_ptr(), B, S, P, L.cin_k, 0, wf.data_ptr(),
                                          bias.data_ptr(), L.k, L.rate, L.pad_b, L.cin_k, L.cout, z2.data_ptr(), L.cout, 0, 0, None, ns, st))
            err = float((z2 - zref).abs().max() / max(1e-30, float(zref.abs().max())))
            row += " x%d v%d %6.3f ms %6.1f TF err %.1e" % (3 if ns == 2 else 6, variant, ms, fl / ms / 1e9, err)
            tot["x%d v%d" % (ns, variant)] = tot.get("x%d v%d" % (ns, variant), 0) + ms
            # filter gradient on the same operands (x terms, and a gradient slab of the output's shape)
            ng = B * (S + 2 * P) ** 2 * L.cout
            g = torch.zeros(ng, device=DEV) if zero else torch.randn(ng, device=DEV)
            gp = torch.zeros(ns * ng, dtype=torch.int16, device=DEV)
            _lib.call("drs_split_terms", g.data_ptr(), ng, ns, gp.data_ptr(), st)
            nsp = _lib.query("drs_conv_wgrad_split_splits", B, S, L.k, L.cin_k, L.cout, P, ns)
            slab = torch.zeros(nsp * w.numel(), device=DEV)
            gw = torch.zeros(w.numel(), device=DEV)
            ms = timeit(lambda: _lib.call("drs_conv_wgrad_split", xp.data_ptr(), B, S, P, L.cin_k, 0, gp.data_ptr(), P, L.cout, 0, L.k, L.rate,
                                          L.pad_b, L.cin_k, L.cin_k, L.cout, slab.data_ptr(), gw.data_ptr(), ns, st))
            row += " wg %6.3f ms %6.1f TF |" % (ms, fl / ms / 1e9)
            tot["wg x%d v%d" % (ns, variant)] = tot.get("wg x%d v%d" % (ns, variant), 0) + ms
            del g, gp, slab
        print(row, flush=True)
    print("total ms:", {k: round(v, 2) for k, v in tot.items()})


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    main(int(kw.get("B", 128)), int(kw.get("S", 64)), kw.get("net", "dilated_grsl_rate8"), kw.get("layers"))
