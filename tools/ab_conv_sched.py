#!/usr/bin/env python3
"""In-process A/B of the launch SCHEDULE of the forward / input-gradient convolution (drs_conv_forward_ws) on the layer shapes of a
net at a per-rank batch (r05): one workgroup per tile, the r03 stream-K cut of every tile, the hybrid (whole tiles for the full rounds,
only the remainder cut), the order of a hybrid workgroup's two parts, wave priority by remaining work.  Interleaved repetitions,
minimum of the medians.  Development aid (libdrs_hip_dev.so).

    python tools/ab_conv_sched.py B=16 S=64,65,85 [arms=plain,sk,hybrid,hybrid_o1,hybrid_o2,prio] [net=dilated_grsl_rate8]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drs_amd import _lib  # noqa: E402
from drs_amd.nets import Plan  # noqa: E402

DEV = "cuda:0"
# arm -> (splitk, hybrid, sk_order, prio)
ARMS = {"auto": (-1, 1, 1, -1), "plain": (0, 1, 0, 0), "sk": (-1, 0, 0, 0), "hybrid": (-1, 1, 0, 0), "hybrid_o1": (-1, 1, 1, 0), "hybrid_o2": (-1, 1, 2, 0),
        "prio": (-1, 1, 0, 1), "plain_prio": (0, 1, 0, 1), "hybrid_o2_prio": (-1, 1, 2, 1), "hybrid_o1_prio": (-1, 1, 1, 1), "sk_prio": (-1, 0, 0, 1)}


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


def setarm(lib, arm):
    sk, hy, od, pr = ARMS[arm]
    lib.drs_debug_conv_splitk(sk)
    lib.drs_debug_conv_hybrid(hy)
    lib.drs_debug_conv_sk_order(od)
    lib.drs_debug_conv_prio(pr)


def main(B, sizes, arms, net):
    lib = _lib.dev()
    plan = Plan(net, 5, 6, first_cin_pad=8)
    st = torch.cuda.current_stream(DEV).cuda_stream
    import ctypes
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
                        setarm(lib, a)
                        res[a].append(timeit(f))
                setarm(lib, "auto")
                bn = 192 if (cout % 192 == 0 and cout % 128) else (128 if cout % 128 == 0 else 64)
                g3 = (ctypes.c_int * 3)()
                lib.drs_debug_conv_sk_geometry(-(-M // 128) * (cout // bn), L.k * L.k * (cin // 32), bn, g3)
                fl = 2.0 * M * L.k * L.k * cin * cout
                best = min(arms, key=lambda a: min(res[a]))
                print("%-6s %-5s k%d r%d %3d->%3d G/W/T=%4d/%4d/%4d " % (L.name, which, L.k, L.rate, cin, cout, g3[0], g3[1], g3[2]) +
                      "  ".join("%s %.3f (%5.1f)" % (a, min(res[a]), fl / min(res[a]) / 1e9) for a in arms) + "   best: " + best, flush=True)
                for a in arms:
                    tot[a] += min(res[a])
        print("total ms: " + "  ".join("%s %.3f" % (a, tot[a]) for a in arms), flush=True)


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    main(int(kw.get("B", 16)), [int(v) for v in kw.get("S", "64,65,85").split(",")], kw.get("arms", "plain,sk,hybrid,hybrid_o1,hybrid_o2,prio").split(","),
         kw.get("net", "dilated_grsl_rate8"))
