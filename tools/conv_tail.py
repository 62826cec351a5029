#!/usr/bin/env python3
"""Where the workgroups of a forward-convolution launch start and end (development build: drs_debug_conv_trace stamps every workgroup
on the 100 MHz real-time clock): how long the launch is, how much of it the chip's workgroup slots are full, how long its tail is.
    python tools/conv_tail.py [B=128] [S=64] [layers=3,8] [skip=1]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drs_amd import _lib
_lib = _lib.dev()
from drs_amd.nets import Plan
DEV = "cuda:0"


def main(B=128, S=64, layers="3,8", skip=1, lpt=1, which="fwd"):
    L_ = _lib.load()
    L_.drs_debug_skip_taps(skip)
    L_.drs_debug_conv_lpt(lpt)
    plan = Plan("dilated_grsl_rate8", 5, 6)
    st = torch.cuda.current_stream(DEV).cuda_stream
    M = B * S * S
    for i, L in enumerate(plan.layers):
        if str(i + 1) not in layers.split(",") or i == 0:
            continue
        P = L.halo
        x = torch.randn(B * (S + 2 * P) ** 2 * L.cin_k, device=DEV)
        w = torch.randn(L.k * L.k * L.cin_k * L.cout, device=DEV) * 0.05
        bias = torch.zeros(L.cout, device=DEV)
        z = torch.zeros(M * L.cout, device=DEV)
        mt = _lib.query("drs_conv_mtile", L.cout)
        stats = torch.zeros(((M + mt - 1) // mt) * L.cout * 2, device=DEV)
        nwg = 16384
        trace = torch.zeros(nwg * 2, dtype=torch.int64, device=DEV)
        f = lambda: _lib.call("drs_conv_forward", x.data_ptr(), B, S, P, L.cin_k, 0, w.data_ptr(), bias.data_ptr(), L.k, L.rate, L.pad_b, L.cin_k,
                              L.cout, z.data_ptr(), L.cout, 0, 0, stats.data_ptr(), st)
        if which == "wgrad":
            g = torch.randn(B * (S + 2 * P) ** 2 * L.cout, device=DEV)
            ns = _lib.query("drs_conv_wgrad_splits", B, S, L.k, L.cin_k, L.cout)
            slab = torch.zeros(ns * L.k * L.k * L.cin_k * L.cout, device=DEV)
            gw = torch.zeros(L.k * L.k * L.cin_k * L.cout, device=DEV)
            f = lambda: _lib.call("drs_conv_wgrad", x.data_ptr(), B, S, P, L.cin_k, 0, g.data_ptr(), P, L.cout, 0, L.k, L.rate, L.pad_b, L.cin_k,
                                  L.cin_k, L.cout, slab.data_ptr(), gw.data_ptr(), st)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        L_.drs_debug_conv_trace(trace.data_ptr())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record()
        torch.cuda.synchronize()
        L_.drs_debug_conv_trace(None)
        t = trace.cpu().numpy().reshape(-1, 2)
        t = t[t[:, 0] > 0]
        t0 = t[:, 0].min()
        s_, e_ = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0          # microseconds
        span = e_.max()
        # slots busy over time
        ev = np.concatenate([np.stack([s_, np.ones_like(s_)], 1), np.stack([e_, -np.ones_like(e_)], 1)])
        ev = ev[np.argsort(ev[:, 0], kind="stable")]
        busy = np.cumsum(ev[:, 1])
        dt = np.diff(np.concatenate([ev[:, 0], [span]]))
        peak = busy.max()
        full = dt[busy >= 0.98 * peak].sum()
        dur = e_ - s_
        tail = span - np.percentile(e_, 100.0 * (1 - peak / len(t)))      # from when the LAST round's first workgroup ends ... roughly: time after which slots only drain
        last_start = s_.max()
        print("%s: %d workgroups, launch %.1f us (event %.1f), peak %d resident; slots >= 98%% full for %.1f us (%.1f %%); last workgroup starts at %.1f us "
              "(drain %.1f us = %.1f %%); workgroup duration min %.1f / median %.1f / max %.1f us; first-round starts spread %.1f us"
              % (L.name, len(t), span, 1e3 * e0.elapsed_time(e1), int(peak), full, 100 * full / span, last_start, span - last_start,
                 100 * (span - last_start) / span, dur.min(), np.median(dur), dur.max(), np.sort(s_)[int(peak) - 1] if peak <= len(s_) else -1), flush=True)
        # area lost: integral of (peak - busy) dt over the launch, as a share of peak * span
        lost = ((peak - busy) * dt).sum() / (peak * span)
        print("      idle slot-time: %.1f %% of the launch (ramp + tail)" % (100 * lost), flush=True)


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    main(int(kw.get("B", 128)), int(kw.get("S", 64)), kw.get("layers", "3,8"), int(kw.get("skip", 1)), int(kw.get("lpt", 1)), kw.get("which", "fwd"))
