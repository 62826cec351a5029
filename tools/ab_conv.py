#!/usr/bin/env python3
"""In-process A/B of the forward / input-gradient kernel on the Dilated8Pooling shapes (development aid): interleaved repetitions on
one device, best-of per arm, bitwise comparison of the outputs.  mode=variant: kernel forms (0 = register-staged, 1 = LDS-DMA);
mode=wide192: Cout = 192 as three 128x64 tiles (0) or one 128x192 tile (1); mode=skip: all-halo tap rows skipped by the size rule (0) or always (1).
    python tools/ab_conv.py [B=128] [S=64] [mode=variant]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drs_amd import _lib
_lib = _lib.dev()      # libdrs_hip_dev.so: the library with the A/B switches of include/drs_dev.h
from drs_amd.nets import Plan
DEV = "cuda:0"

def main(B=128, S=64, rounds=4, mode="variant"):
    lib = _lib.load()
    setter = {"variant": lib.drs_debug_conv_variant, "wide192": lib.drs_debug_conv_wide192, "skip": (lambda v: lib.drs_debug_skip_taps(1 + v))}[mode]
    plan = Plan("dilated_grsl_rate8", 5, 6, first_cin_pad=8)
    st = torch.cuda.current_stream(DEV).cuda_stream
    M = B * S * S
    tot = {(v, d): 0.0 for v in (0, 1) for d in ("fwd", "dgrad")}
    for li, L in enumerate(plan.layers):
        if li == 0:
            continue
        P = L.halo
        x = torch.randn(B * (S + 2 * P) ** 2 * L.cin_k, device=DEV)
        g = torch.randn(B * (S + 2 * P) ** 2 * L.cout, device=DEV)
        w = torch.randn(L.k * L.k * L.cin_k * L.cout, device=DEV) * 0.05
        bias = torch.randn(L.cout, device=DEV)
        mt = _lib.query("drs_conv_mtile", L.cout)
        outs = {}
        best = {k: 1e9 for k in tot}
        row = "%-6s" % L.name
        for r in range(rounds):
            for v in (0, 1):
                setter(v)
                for d in ("fwd", "dgrad"):
                    cin, cout, inp, pad = (L.cin_k, L.cout, x, L.pad_b) if d == "fwd" else (L.cout, L.cin_k, g, L.pad_a)
                    z = torch.zeros(M * cout, device=DEV)
                    stats = torch.zeros(((M + mt - 1) // mt) * cout * 2, device=DEV) if d == "fwd" else None
                    for rep in range(3):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        _lib.call("drs_conv_forward", inp.data_ptr(), B, S, P, cin, 0, w.data_ptr(), bias.data_ptr() if d == "fwd" else None, L.k, L.rate,
                                  pad, cin, cout, z.data_ptr(), cout, 0, 0, stats.data_ptr() if stats is not None else None, st)
                        e1.record()
                        torch.cuda.synchronize()
                        if rep:
                            best[(v, d)] = min(best[(v, d)], e0.elapsed_time(e1))
                    if r == 0:
                        outs[(v, d)] = (z.clone(), None if stats is None else stats.clone())
        fl = 2.0 * M * L.k * L.k * L.cin_k * L.cout
        same = all(torch.equal(outs[(0, d)][0], outs[(1, d)][0]) for d in ("fwd", "dgrad")) and torch.equal(outs[(0, "fwd")][1], outs[(1, "fwd")][1])
        for d in ("fwd", "dgrad"):
            row += "  %s: v0 %6.3f ms %5.1f TF | v1 %6.3f ms %5.1f TF (%+.1f %%) |" % (d, best[(0, d)], fl / best[(0, d)] / 1e9, best[(1, d)],
                                                                                   fl / best[(1, d)] / 1e9, 100 * (best[(1, d)] / best[(0, d)] - 1))
            for v in (0, 1):
                tot[(v, d)] += best[(v, d)]
        print(row + ("  bitwise equal" if same else "  OUTPUTS DIFFER"), flush=True)
    print("total  " + "  ".join("%s v%d %.3f ms" % (d, v, tot[(v, d)]) for d in ("fwd", "dgrad") for v in (0, 1)))
    lib.drs_debug_conv_variant(-1)
    lib.drs_debug_conv_wide192(1)
    lib.drs_debug_skip_taps(1)

if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    main(int(kw.get("B", 128)), int(kw.get("S", 64)), int(kw.get("rounds", 4)), kw.get("mode", "variant"))
