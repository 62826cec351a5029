#!/usr/bin/env python3
"""In-process A/B of the launch order of the plain forward / input-gradient launches, per layer of Dilated8Pooling, forward and input
gradient, interleaved, best of.  Arms (drs_debug_skip_taps, drs_debug_conv_lpt):
    old   (1, 0)  halo-tap rows skipped from 4096 workgroups, natural order (r03)
    new   (1, 1)  full tiles first, the skipping ones last, skipping wherever that order applies (default)
    all   (0, 0)  every tap multiplied
    python tools/ab_lpt.py [B=128] [S=64] [rounds=5] [arms=old,new,all]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drs_amd import _lib
_lib = _lib.dev()
from drs_amd.nets import Plan
DEV = "cuda:0"
ARMS = {"old": (1, 0), "new": (1, 1), "all": (0, 0)}


def main(B=128, S=64, rounds=5, arms="old,new,all"):
    L_ = _lib.load()
    arms = arms.split(",")
    plan = Plan("dilated_grsl_rate8", 5, 6)
    st = torch.cuda.current_stream(DEV).cuda_stream
    M = B * S * S
    tot = {(d, a): 0.0 for d in ("fwd", "dgrad") for a in arms}
    ws_n = max(_lib.query("drs_conv_workspace_floats", c) for c in (64, 128, 192, 256))
    ws = torch.zeros(ws_n, device=DEV)
    for i, L in enumerate(plan.layers):
        if i == 0:
            continue
        P = L.halo
        x = torch.randn(B * (S + 2 * P) ** 2 * L.cin_k, device=DEV)
        g = torch.randn(B * (S + 2 * P) ** 2 * L.cout, device=DEV)
        w = torch.randn(L.k * L.k * L.cin_k * L.cout, device=DEV) * 0.05
        bias = torch.zeros(L.cout, device=DEV)
        z = torch.zeros(M * max(L.cout, L.cin_k), device=DEV)
        mt = _lib.query("drs_conv_mtile", L.cout)
        stats = torch.zeros(((M + mt - 1) // mt) * L.cout * 2, device=DEV)
        fns = {"fwd": lambda: _lib.call("drs_conv_forward_ws", x.data_ptr(), B, S, P, L.cin_k, 0, w.data_ptr(), bias.data_ptr(), L.k, L.rate, L.pad_b,
                                        L.cin_k, L.cout, z.data_ptr(), L.cout, 0, 0, stats.data_ptr(), ws.data_ptr(), ws_n, st),
               "dgrad": lambda: _lib.call("drs_conv_forward_ws", g.data_ptr(), B, S, P, L.cout, 0, w.data_ptr(), None, L.k, L.rate, L.pad_a, L.cout,
                                          L.cin_k, z.data_ptr(), L.cin_k, 0, 0, None, ws.data_ptr(), ws_n, st)}
        row = "%-6s" % L.name
        for d, f in fns.items():
            best = {a: 1e9 for a in arms}
            for r in range(rounds):
                for a in arms:
                    L_.drs_debug_skip_taps(ARMS[a][0])
                    L_.drs_debug_conv_lpt(ARMS[a][1])
                    for rep in range(3):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(); f(); e1.record()
                        torch.cuda.synchronize()
                        if rep:
                            best[a] = min(best[a], e0.elapsed_time(e1))
            row += "  %s " % d + " | ".join("%s %.3f" % (a, best[a]) for a in arms) + " ms"
            for a in arms:
                tot[(d, a)] += best[a]
        print(row, flush=True)
    L_.drs_debug_conv_lpt(1)
    L_.drs_debug_skip_taps(1)
    print("total  fwd " + " | ".join("%s %.3f" % (a, tot[("fwd", a)]) for a in arms) + " ms   dgrad " + " | ".join("%s %.3f" % (a, tot[("dgrad", a)]) for a in arms) + " ms")


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    main(int(kw.get("B", 128)), int(kw.get("S", 64)), int(kw.get("rounds", 5)), kw.get("arms", "old,new,all"))
