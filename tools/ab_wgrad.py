#!/usr/bin/env python3
"""In-process A/B of the filter-gradient launch policy on the Dilated8Pooling shapes (development aid): interleaved repetitions on
one device, best-of per arm.  An arm is a string of development-switch settings: b<0|1> cut by live pixels off / on
(drs_debug_wgrad_balance), t<N> workgroup target (drs_debug_wgrad_target), g<N> target of the many-tiles-and-pixels launches under
the live cut (drs_debug_wgrad_target_big), v<0|1> kernel form (drs_debug_wgrad_variant; default per tile), l<N> chunks per workgroup that small launches aim at,
m<N> fewest chunks a split may have (drs_debug_wgrad_minchunks), o<0|1> workgroup count by the r02 table / the per-CU cost model,
a<0|1> timing experiment with WRONG sums: every tap reads the un-shifted pixels (what perfect re-use of X across tap rows would buy),
s<0|1> sides >= 32 that are not a multiple of 32 in the LDS-DMA form: the r04 offset tables / row-segment addressing (drs_debug_wgrad_seg).
S may be a list (S=63,64,65): one table per side with ms per 10^6 pixels, for comparing sides at equal pixels.
    python tools/ab_wgrad.py [B=128] [S=64] [arms=b0,b1,b1g2048] [layers=1,2,...] [rounds=4]"""
import os, re, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drs_amd import _lib
_lib = _lib.dev()      # libdrs_hip_dev.so: the library with the A/B switches of include/drs_dev.h
from drs_amd.nets import Plan
DEV = "cuda:0"


def apply(lib, arm):
    kv = dict((m.group(1), int(m.group(2))) for m in re.finditer(r"([btgvlmoasp])(\d+)", arm))
    lib.drs_debug_wgrad_balance(kv.get("b", 1))
    lib.drs_debug_wgrad_target(kv.get("t", 2048))
    lib.drs_debug_wgrad_target_big(kv.get("g", 0))
    lib.drs_debug_wgrad_variant(kv.get("v", -1))
    lib.drs_debug_wgrad_len(kv.get("l", 0))
    lib.drs_debug_wgrad_minchunks(kv.get("m", 8))
    lib.drs_debug_wgrad_model(kv.get("o", 1))
    lib.drs_debug_wgrad_ablate(kv.get("a", 0))
    lib.drs_debug_wgrad_seg(kv.get("s", 1))
    lib.drs_debug_wgrad_prio(kv.get("p", -1))      # p<0|1|2>: wave priority by remaining work never / levels 3..0 / levels 2..0 (default: by the rule)


def main(B=128, S=64, arms="b0,b1", rounds=4, layers=""):
    lib = _lib.load()
    plan = Plan("dilated_grsl_rate8", 5, 6, first_cin_pad=8)
    st = torch.cuda.current_stream(DEV).cuda_stream
    arms = arms.split(",")
    tot = {a: 0.0 for a in arms}
    for li, L in enumerate(plan.layers):
        if layers and str(li + 1) not in layers.split(","):
            continue
        P = L.halo
        x = torch.randn(B * (S + 2 * P) ** 2 * L.cin_k, device=DEV)
        g = torch.randn(B * (S + 2 * P) ** 2 * L.cout, device=DEV)
        gw = torch.zeros(L.k * L.k * L.cin_k * L.cout, device=DEV)
        best = {a: 1e9 for a in arms}
        slabs, nwg = {}, {}
        for a in arms:
            apply(lib, a)
            ns = _lib.query("drs_conv_wgrad_splits", B, S, L.k, L.cin_k, L.cout)
            slabs[a] = torch.zeros(ns * L.k * L.k * L.cin_k * L.cout, device=DEV)
            nwg[a] = lib.drs_debug_wgrad_cut(B, S, L.k, L.rate, L.pad_b, L.cin_k, L.cout, None, 0, None, None)
        for r in range(rounds):
            for a in arms:
                apply(lib, a)
                for rep in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    _lib.call("drs_conv_wgrad", x.data_ptr(), B, S, P, L.cin_k, 0, g.data_ptr(), P, L.cout, 0, L.k, L.rate, L.pad_b, L.cin_k,
                              L.cin_k, L.cout, slabs[a].data_ptr(), gw.data_ptr(), st)
                    e1.record()
                    torch.cuda.synchronize()
                    if rep:
                        best[a] = min(best[a], e0.elapsed_time(e1))
        fl = 2.0 * B * S * S * L.k * L.k * L.cin * L.cout
        print("%-6s" % L.name + "".join("  %s (%4d wg) %6.3f ms %5.1f TF |" % (a, nwg[a], best[a], fl / best[a] / 1e9) for a in arms), flush=True)
        for a in arms:
            tot[a] += best[a]
        del slabs
    print("total  " + "".join("  %s %7.3f ms (%6.3f ms per Mpx) |" % (a, tot[a], tot[a] / (B * S * S / 1e6)) for a in arms), flush=True)
    apply(lib, "")


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    for S_ in kw.get("S", "64").split(","):
        print("== B=%s S=%s" % (kw.get("B", 128), S_), flush=True)
        main(int(kw.get("B", 128)), int(S_), kw.get("arms", "b0,b1"), int(kw.get("rounds", 4)), kw.get("layers", ""))
