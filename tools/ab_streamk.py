#!/usr/bin/env python3
"""In-process A/B of the stream-K cut of the forward / input-gradient convolution (drs_conv_forward_ws) on the layer shapes of a
net at a per-rank batch: one workgroup per tile (off), the library's rule (auto) and forced workgroup counts, interleaved
repetitions, minimum of the medians.  Development aid (libdrs_hip_dev.so).

    python tools/ab_streamk.py B=16 S=25,35,45,55,65,75,85 [W=256,512,768] [net=dilated_grsl_rate8]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drs_amd import _lib  # noqa: E402
from drs_amd.nets import Plan  # noqa: E402

DEV = "cuda:0"


def timeit(fn, reps=5):
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


def main(B, sizes, Ws, net):
    lib = _lib.dev()
    plan = Plan(net, 5, 6, first_cin_pad=8)
    st = torch.cuda.current_stream(DEV).cuda_stream
    arms = ["off", "auto"] + [str(w) for w in Ws]
    for S in sizes:
        M = B * S * S
        tot = {a: 0.0 for a in arms}
        print("== B=%d S=%d (M=%d, %d M tiles)" % (B, S, M, -(-M // 128)))
        for i, L in enumerate(plan.layers):
            if L.cin_k < 32:
                continue
            for which in ("fwd", "dgrad"):
                cin, cout, pad = (L.cin_k, L.cout, L.pad_b) if which == "fwd" else (L.cout, L.cin_k, L.pad_a)
                P = L.halo
                x = torch.randn(B * (S + 2 * P) ** 2 * cin, device=DEV)
                w = torch.randn(L.k * L.k * cin * cout, device=DEV) * 0.05
                z = torch.zeros(M * cout, device=DEV)
                nws = lib.query("drs_conv_workspace_floats", cout)
                ws = torch.zeros(max(nws, 1), device=DEV)
                mt = lib.query("drs_conv_mtile", cout)
                stats = torch.zeros(((M + mt - 1) // mt) * cout * 2, device=DEV) if which == "fwd" else None
                f = lambda: lib.call("drs_conv_forward_ws", x.data_ptr(), B, S, P, cin, 0, w.data_ptr(), None, L.k, L.rate, pad, cin, cout, z.data_ptr(),
                                     cout, 0, 0, stats.data_ptr() if stats is not None else None, ws.data_ptr(), nws, st)
                res = {a: [] for a in arms}
                for rep in range(3):
                    for a in arms:
                        lib.drs_debug_conv_splitk({"off": 0, "auto": -1}.get(a, int(a) if a.isdigit() else 0))
                        res[a].append(timeit(f))
                lib.drs_debug_conv_splitk(-1)
                fl = 2.0 * M * L.k * L.k * cin * cout
                best = min(arms, key=lambda a: min(res[a]))
                print("%-6s %-5s k%d r%d %3d->%3d  " % (L.name, which, L.k, L.rate, cin, cout) +
                      "  ".join("%s %.3f ms (%5.1f TF)" % (a, min(res[a]), fl / min(res[a]) / 1e9) for a in arms) + "   best: " + best, flush=True)
                for a in arms:
                    tot[a] += min(res[a])
        print("total ms: " + "  ".join("%s %.3f" % (a, tot[a]) for a in arms), flush=True)


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    main(int(kw.get("B", 16)), [int(v) for v in kw.get("S", "25,45,65").split(",")], [int(v) for v in kw.get("W", "256,512,768").split(",")],
         kw.get("net", "dilated_grsl_rate8"))
