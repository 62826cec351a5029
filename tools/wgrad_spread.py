#!/usr/bin/env python3
"""Why do the equal-length workgroups of a filter-gradient launch differ in duration?  Per-workgroup (start, end) stamps of the
development build against the host's copy of the cut (row tile, column tile, split, chunks) and the XCD a workgroup ran on."""
import os, sys
import ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drs_amd import _lib
_lib = _lib.dev()
from drs_amd.nets import Plan
DEV = "cuda:0"


def main(B=128, S=64, layers="8"):
    L_ = _lib.load()
    plan = Plan("dilated_grsl_rate8", 5, 6)
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
        f = lambda: _lib.call("drs_conv_wgrad", x.data_ptr(), B, S, P, L.cin_k, 0, g.data_ptr(), P, L.cout, 0, L.k, L.rate, L.pad_b, L.cin_k,
                              L.cin_k, L.cout, slab.data_ptr(), gw.data_ptr(), st)
        n = L_.drs_debug_wgrad_cut(B, S, L.k, L.rate, L.pad_b, L.cin_k, L.cout, None, 0, None, None)
        cut = (C.c_int * (5 * n))()
        L_.drs_debug_wgrad_cut(B, S, L.k, L.rate, L.pad_b, L.cin_k, L.cout, cut, n, None, None)
        cut = np.asarray(cut).reshape(n, 5)          # (row tile, column tile, split, first chunk, end chunk) of LOGICAL workgroup i
        trace = torch.zeros(3 * 16384, dtype=torch.int64, device=DEV)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        L_.drs_debug_conv_trace(trace.data_ptr())
        f()
        torch.cuda.synchronize()
        L_.drs_debug_conv_trace(None)
        hw = trace.cpu().numpy()[32768:32768 + n]
        t = trace.cpu().numpy()[:32768].reshape(-1, 2)[:n]
        t0 = t[:, 0].min()
        s_, e_ = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0
        dur = e_ - s_
        bid = np.arange(n)
        xcd = bid & 7
        # logical index of block b (xcd_remap)
        q, r = n >> 3, n & 7
        start = np.where(xcd < r, xcd * (q + 1), r * (q + 1) + (xcd - r) * q)
        logical = start + (bid >> 3)
        rt, ct, sp, c0, c1 = cut[logical].T
        print("%s: %d workgroups, launch %.0f us; duration mean %.0f, std %.0f (%.1f %%), min %.0f, max %.0f" % (L.name, n, e_.max(), dur.mean(), dur.std(), 100 * dur.std() / dur.mean(), dur.min(), dur.max()))
        print("   by XCD (mean us):", " ".join("%.0f" % dur[xcd == k].mean() for k in range(8)))
        rounds = np.minimum((s_ // (dur.mean() * 0.9)).astype(int), 5)
        print("   by start time (round: count, mean duration):", " ".join("%d: %d, %.0f |" % (k, (rounds == k).sum(), dur[rounds == k].mean()) for k in range(rounds.max() + 1)))
        print("   by row tile (mean us):", " ".join("%.0f" % dur[rt == k].mean() for k in range(rt.max() + 1)))
        print("   chunks per workgroup: min %d max %d; corr(duration, chunks) %.2f; corr(duration, split index) %.2f" % ((c1 - c0).min(), (c1 - c0).max(), np.corrcoef(dur, c1 - c0)[0, 1] if (c1 - c0).std() > 0 else 0, np.corrcoef(dur, sp)[0, 1]))
        hwid, xcc = hw & 0xffffffff, (hw >> 32) & 0xf
        cu = (xcc << 8) | (((hwid >> 13) & 7) << 5) | (((hwid >> 12) & 1) << 4) | ((hwid >> 8) & 15)       # (xcc, se, sh, cu)
        firstr = s_ < 5.0
        ids = np.unique(cu[firstr])
        within = np.mean([dur[firstr & (cu == c)].std() for c in ids if (firstr & (cu == c)).sum() > 1])
        means = np.array([dur[firstr & (cu == c)].mean() for c in ids])
        print("   first round by CU: %d distinct CUs, workgroups per CU %s; std of a CU's workgroups %.0f us, std of the CU means %.0f us (min %.0f max %.0f)"
              % (len(ids), np.bincount(np.bincount(np.searchsorted(ids, cu[firstr]))).tolist(), within, means.std(), means.min(), means.max()))
        print("   CU mean duration by XCC:", " ".join("%.0f" % means[(ids >> 8) == k].mean() for k in range(8)), "| by SE:", " ".join("%.0f" % means[((ids >> 5) & 7) == k].mean() for k in range(4)))
        first = s_ < 5.0
        print("   first round only (%d workgroups start within 5 us): duration mean %.0f std %.0f min %.0f max %.0f; their end times span %.0f .. %.0f us" % (first.sum(), dur[first].mean(), dur[first].std(), dur[first].min(), dur[first].max(), e_[first].min(), e_[first].max()))


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    main(int(kw.get("B", 128)), int(kw.get("S", 64)), kw.get("layers", "8"))
