#!/usr/bin/env python3
"""Time the training step at a given local batch (what one rank sees under data parallelism).  Default: no collectives;
`comm=rccl`: every collective of the step issued at world 1 (identities; DRS_FORCE_COLLECTIVES) by the LIBRARY itself through RCCL
(drs_net_set_rccl: no Python in the step); `comm=callback`: the same sums through the all-reduce callback into torch.distributed
(DRS_COMM=torch; the r02 path); `comm=callback2`: that with torch's second communicator for the small sums (DRS_BN_COMM)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drs_amd.net import DilatedNet, KernelTimer
from drs_amd import patches as P
from drs_amd.synthetic import make_tile, grid_instances

def main(B=16, S=64, steps=20, arith="f32", comm_kind="none", two=None):
    dev = "cuda:0"
    comm = None
    if comm_kind != "none":
        os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29534", DRS_FORCE_COLLECTIVES="1")
        os.environ["DRS_COMM"] = "rccl" if comm_kind == "rccl" else "torch"
        if os.environ.get("BENCH_RCCL_LIB"):      # this tool's own variable (a stand-in whose sums are real launches, tools/ubench/nccl_latency_double.hip)
            from drs_amd import _lib
            _lib.call("drs_rccl_bind_library", os.environ["BENCH_RCCL_LIB"].encode())
        if comm_kind == "callback2":
            os.environ["DRS_BN_COMM"] = "1"
        from drs_amd.dist import TorchComm
        torch.cuda.set_device(0)
        comm = TorchComm("nccl")
    tile, lab = make_tile(1024, 1024, 5, 6, seed=1234)
    pool = P.TilePool([tile], [lab], dev)
    inst = grid_instances(1024, 1024, S, 25, 4096, seed=0)
    net = DilatedNet("dilated_grsl_rate8", 5, 6, 0.005, b_max=B, s_max=S, device=dev, arith=arith, comm=comm)
    if two is not None:          # two=0|1: the backward pass on one / two streams whatever the library's rule says (drs_net_set_two_streams)
        net.set_two_streams(two)
    np.random.seed(0)
    def step(i):
        rows = inst[(i * B) % 4000:(i * B) % 4000 + B]
        aug = P.draw_augmentation(rows, S, 5, noise="device")
        P.crop_to_net(net, pool, rows, S, [0.5] * 3, [0.2] * 3, aug)
        return net.train_step(B, S, 0.01)
    for i in range(5): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps): step(i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    # host-only time per step (enqueue cost): run without waiting, measure enqueue loop
    t1 = time.perf_counter()
    for i in range(steps): step(i)
    host = (time.perf_counter() - t1) / steps
    torch.cuda.synchronize()
    net.timer = KernelTimer()
    for i in range(3): step(i)
    summ = net.timer.summary(); net.timer = None
    ksum = sum(d["ms"] for d in summ.values()) / 3
    print(arith, "comm=" + comm_kind + ("/" + str(getattr(net, "collectives", None)) if comm else ""), "B=%d S=%d: %.2f ms/step  (%.0f patches/s; x%d ranks = %.0f)  host enqueue %.2f ms/step, timed kernels %.2f ms" % (B, S, dt * 1e3, B / dt, 128 // B, 128 / dt, host * 1e3, ksum))
    for k, d in sorted(summ.items()): print("   %-18s %6.3f ms/step" % (k, d["ms"] / 3))

if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    if "sched" in kw:          # r06 experiment: the filter gradients' schedule (tools/ab_wgrad_schedule.py: p | L<streams><e|d|o>), the whole net on the dev library
        import re
        from drs_amd import _lib
        d = _lib.dev()
        _lib._lib = d.lib
        m = re.fullmatch(r"L(\d)([edoa])", kw["sched"])
        if m:
            d.drs_debug_wgrad_schedule(1, int(m.group(1)), {"e": 0, "d": 1, "o": 2, "a": 3}[m.group(2)])
    if "slide" in kw or "minrows" in kw:        # A/B of the sliding elementwise kernels' launch geometry inside the step: the whole net on the dev library
        from drs_amd import _lib
        d = _lib.dev()
        _lib._lib = d.lib
        if "slide" in kw: d.drs_debug_slide_blocks(int(kw["slide"]))
        if "minrows" in kw: d.drs_debug_slide_minrows(int(kw["minrows"]))
    main(int(kw.get("B", 16)), int(kw.get("S", 64)), int(kw.get("steps", 20)), kw.get("arith", "f32"), kw.get("comm", "none"),
         int(kw["two"]) if "two" in kw else None)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
