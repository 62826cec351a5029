#!/usr/bin/env python3
"""In-process A/B of the filter-gradient kernel's split targets (the variant / stagger arms are placeholders of past experiments, profiles/r02/wgrad_ablation.txt) on the Dilated8Pooling shapes (development aid):
interleaved repetitions on one device, best-of per arm.   python tools/ab_wgrad.py [B=128] [S=64] [targets=1536,2048] [variants=0,1]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drs_amd import _lib
from drs_amd.nets import Plan
DEV = "cuda:0"

def main(B=128, S=64, targets="1536", variants="0,1", rounds=4, staggers="0", layers=""):
    lib = _lib.load()
    plan = Plan("dilated_grsl_rate8", 5, 6, first_cin_pad=8)
    st = torch.cuda.current_stream(DEV).cuda_stream
    arms = [(int(v), int(t), int(g)) for v in variants.split(",") for t in targets.split(",") for g in staggers.split(",")]
    tot = {a: 0.0 for a in arms}
    for li, L in enumerate(plan.layers):
        if layers and str(li + 1) not in layers.split(","):
            continue
        P = L.halo
        x = torch.randn(B * (S + 2 * P) ** 2 * L.cin_k, device=DEV)
        g = torch.randn(B * (S + 2 * P) ** 2 * L.cout, device=DEV)
        gw = torch.zeros(L.k * L.k * L.cin_k * L.cout, device=DEV)
        best = {a: 1e9 for a in arms}
        slabs = {}
        for (v, tg, sg) in arms:
            lib.drs_debug_wgrad_target(tg)
            ns = _lib.query("drs_conv_wgrad_splits", B, S, L.k, L.cin_k, L.cout)
            slabs[(v, tg, sg)] = (ns, torch.zeros(ns * L.k * L.k * L.cin_k * L.cout, device=DEV))
        for r in range(rounds):
            for (v, tg, sg) in arms:
                lib.drs_debug_wgrad_variant(v)
                lib.drs_debug_wgrad_target(tg)
                ns, slab = slabs[(v, tg, sg)]
                for rep in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    _lib.call("drs_conv_wgrad", x.data_ptr(), B, S, P, L.cin_k, 0, g.data_ptr(), P, L.cout, 0, L.k, L.rate, L.pad_b, L.cin_k,
                              L.cin_k, L.cout, slab.data_ptr(), gw.data_ptr(), st)
                    e1.record()
                    torch.cuda.synchronize()
                    if rep:
                        best[(v, tg, sg)] = min(best[(v, tg, sg)], e0.elapsed_time(e1))
        fl = 2.0 * B * S * S * L.k * L.k * L.cin * L.cout
        print("%-6s" % L.name + "".join("  v%d t%d s%d (%3d) %6.3f ms %5.1f TF |" % (v, tg, sg, slabs[(v, tg, sg)][0], best[(v, tg, sg)], fl / best[(v, tg, sg)] / 1e9)
                                         for (v, tg, sg) in arms), flush=True)
        for a in arms:
            tot[a] += best[a]
        del slabs
    print("total  " + "".join("  v%d t%d s%d %7.3f ms |" % (v, tg, sg, tot[(v, tg, sg)]) for (v, tg, sg) in arms))
    lib.drs_debug_wgrad_target(2048)

if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    main(int(kw.get("B", 128)), int(kw.get("S", 64)), kw.get("targets", "1536"), kw.get("variants", "0,1"), int(kw.get("rounds", 4)), kw.get("staggers", "0"), kw.get("layers", ""))
