#!/usr/bin/env python3
"""In-process A/B of one rank's training step (development library) over wave-priority / stream arms, interleaved on one device.
An arm is a string of settings: t<-1|0|1> two-stream backward pass (drs_net_set_two_streams: a net per value; 9 = the rule),
w<0|1|2|9> filter gradient's wave priority by remaining work (drs_debug_wgrad_prio; 9 = the rule), c<0|1|3|9> forward /
input-gradient kernel (drs_debug_conv_prio; 3 = every launch at the top level; 9 = the rule),
e<0|1|2|9> the chain of the two-stream backward pass at the top level (drs_debug_chain_mode: 1 input-gradient launches, 2 + batch-norm backward; 9 = as the engine asks: 1 without collectives, 2 with),
r<0|1> the classifier's slab reductions on the filter-gradient stream (0, default) or on the chain as before round 5 (1),
a<0|1> TIMING EXPERIMENT with wrong sums: the filter gradient reads the un-shifted pixels for every tap (what perfect re-use of X would buy).
    python tools/ab_step_prio.py [B=16] [S=64,65] [arms=t9w9c9,t9w2c9,...] [steps=20] [rounds=4] [comm=rccl]"""
import os, re, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drs_amd import _lib
d = _lib.dev()
_lib._lib = d.lib          # the whole net on libdrs_hip_dev.so
from drs_amd.net import DilatedNet
from drs_amd import patches as P
from drs_amd.synthetic import make_tile, grid_instances


def parse(arm):
    kv = dict((m.group(1), int(m.group(2))) for m in re.finditer(r"([twcear])(-?\d+)", arm))
    f = lambda k: -1 if kv.get(k, 9) == 9 else kv[k]
    return f("t"), f("w"), f("c"), (-1 if kv.get("e", 9) == 9 else kv["e"]), kv.get("a", 0), kv.get("r", 0)


def main(B=16, Ss=(64,), arms=("t9w9c9",), steps=20, rounds=4, comm_kind="none"):
    dev = "cuda:0"
    comm = None
    if comm_kind == "rccl":      # every collective of the step issued by the library at world 1 (AB_RCCL_LIB, this tool's variable: through that
        # NCCL-API library, e.g. tools/ubench/nccl_latency_double.hip, whose all-reduces are real launches with a wire time)
        if os.environ.get("AB_RCCL_LIB"):
            _lib.call("drs_rccl_bind_library", os.environ["AB_RCCL_LIB"].encode())
        os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29536", DRS_FORCE_COLLECTIVES="1", DRS_COMM="rccl")
        from drs_amd.dist import TorchComm
        torch.cuda.set_device(0)
        comm = TorchComm("nccl")
    tile, lab = make_tile(1024, 1024, 5, 6, seed=1234)
    pool = P.TilePool([tile], [lab], dev)
    nets = {}
    smax = max(Ss)
    for a in arms:
        t = parse(a)[0]
        if t not in nets:
            nets[t] = DilatedNet("dilated_grsl_rate8", 5, 6, 0.005, b_max=B, s_max=smax, device=dev, comm=comm)
            nets[t].set_two_streams(t)
    for S in Ss:
        inst = grid_instances(1024, 1024, S, 25, 4096, seed=0)
        best = {a: [] for a in arms}
        for r in range(rounds + 1):
            for a in arms:
                t, w, c, e, ab, rc = parse(a)
                net = nets[t]
                d.drs_debug_wgrad_prio(w)
                d.drs_debug_conv_prio(c)
                d.drs_debug_chain_mode(e)
                d.drs_debug_wgrad_ablate(ab)
                d.drs_debug_reductions_on_chain(1 if rc else 0)
                np.random.seed(0)
                def step(i):
                    rows = inst[(i * B) % 4000:(i * B) % 4000 + B]
                    aug = P.draw_augmentation(rows, S, 5, noise="device")
                    P.crop_to_net(net, pool, rows, S, [0.5] * 3, [0.2] * 3, aug)
                    return net.train_step(B, S, 0.01)
                for i in range(3):
                    step(i)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(steps):
                    step(i)
                torch.cuda.synchronize()
                if r:
                    best[a].append((time.perf_counter() - t0) / steps * 1e3)
        print("B=%d S=%d  " % (B, S) + "   ".join("%s %.3f (min %.3f)" % (a, float(np.median(v)), min(v)) for a, v in best.items()), flush=True)
    d.drs_debug_wgrad_prio(-1)
    d.drs_debug_conv_prio(-1)
    d.drs_debug_chain_mode(-1)
    d.drs_debug_wgrad_ablate(0)


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    main(int(kw.get("B", 16)), [int(v) for v in kw.get("S", "64").split(",")], kw.get("arms", "t9w9c9,t9w1c9,t9w2c9,t9w2c3,t0w9c9,t0w1c9").split(","),
         int(kw.get("steps", 20)), int(kw.get("rounds", 4)), kw.get("comm", "none"))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
