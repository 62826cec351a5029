#!/usr/bin/env python3
"""In-process A/B of the launch order of the plain forward / input-gradient launches (drs_debug_conv_lpt): natural order against
"full tiles first, halo-skipping tiles last", per layer of Dilated8Pooling, forward and input gradient, interleaved, best of."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drs_amd import _lib
_lib = _lib.dev()
from drs_amd.nets import Plan
DEV = "cuda:0"


def main(B=128, S=64, rounds=6, force_skip=0):
    L_ = _lib.load()
    plan = Plan("dilated_grsl_rate8", 5, 6)
    st = torch.cuda.current_stream(DEV).cuda_stream
    M = B * S * S
    tot = {(d, v): 0.0 for d in ("fwd", "dgrad") for v in (0, 1)}
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
        fns = {"fwd": lambda: _lib.call("drs_conv_forward", x.data_ptr(), B, S, P, L.cin_k, 0, w.data_ptr(), bias.data_ptr(), L.k, L.rate, L.pad_b,
                                        L.cin_k, L.cout, z.data_ptr(), L.cout, 0, 0, stats.data_ptr(), st),
               "dgrad": lambda: _lib.call("drs_conv_forward", g.data_ptr(), B, S, P, L.cout, 0, w.data_ptr(), None, L.k, L.rate, L.pad_a, L.cout,
                                          L.cin_k, z.data_ptr(), L.cin_k, 0, 0, None, st)}
        row = "%-6s" % L.name
        for d, f in fns.items():
            best = {0: 1e9, 1: 1e9, 2: 1e9}
            for r in range(rounds):
                for v in ((0, 1, 2) if force_skip else (0, 1)):
                    L_.drs_debug_conv_lpt(1 if v == 1 else 0)
                    if force_skip:
                        L_.drs_debug_skip_taps(0 if v == 2 else 2)      # arm 2: every tap multiplied (what small launches do today)
                    for rep in range(3):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(); f(); e1.record()
                        torch.cuda.synchronize()
                        if rep:
                            best[v] = min(best[v], e0.elapsed_time(e1))
            row += "  %s natural %.3f ms | full-first %.3f ms (%+.1f %%)" % (d, best[0], best[1], 100 * (best[1] / best[0] - 1))
            if force_skip:
                row += " | no skipping %.3f ms" % best[2]
            tot[(d, 0)] += best[0]; tot[(d, 1)] += best[1]
        print(row, flush=True)
    L_.drs_debug_conv_lpt(1)
    L_.drs_debug_skip_taps(1)
    print("total  fwd %.3f -> %.3f ms   dgrad %.3f -> %.3f ms" % (tot[("fwd", 0)], tot[("fwd", 1)], tot[("dgrad", 0)], tot[("dgrad", 1)]))


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    main(int(kw.get("B", 128)), int(kw.get("S", 64)), int(kw.get("rounds", 6)), int(kw.get("force_skip", 0)))
