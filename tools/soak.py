#!/usr/bin/env python3
"""Soak / determinism run of the step engine in the per-rank regime: `steps` training steps at batch B with the patch side drawn
uniformly from [lo, hi] per step (isprs:1727-1737), run TWICE from the same seeds; the two runs must agree bit for bit in every
loss and in the final variables (stream-K cuts, the two-stream backward pass, the LDS-exchange BN kernels and the fixed-order
reductions all promise that), and the loss must stay finite.   python tools/soak.py [B=16] [steps=1500] [lo=25] [hi=85] [net=dilated_grsl_rate8 channels=5 classes=6] [comm=none|rccl|callback]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drs_amd.net import DilatedNet
from drs_amd import patches as P
from drs_amd.synthetic import make_tile, grid_instances


def run(B, steps, lo, hi, pool, dev, net_type="dilated_grsl_rate8", channels=5, classes=6, comm=None):
    net = DilatedNet(net_type, channels, classes, 0.005, b_max=B, s_max=hi, device=dev, seed=42, comm=comm)
    rng = np.random.default_rng(7)
    np.random.seed(11)
    inst = {}
    losses = torch.zeros(steps, 2, dtype=torch.float64, device=dev)
    t0 = time.perf_counter()
    for i in range(steps):
        S = int(rng.integers(lo, hi + 1))
        if S not in inst:
            inst[S] = grid_instances(1024, 1024, S, 25, 4096, seed=S)
        rows = inst[S][(i * B) % 4000:(i * B) % 4000 + B]
        aug = P.draw_augmentation(rows, S, channels, noise="device")
        P.crop_to_net(net, pool, rows, S, [0.5] * 3, [0.2] * 3, aug)
        out = net.train_step(B, S, 0.01)
        losses[i] = out["loss_parts"]
        if i and i % 5000 == 0:
            print("   step %d, CE %.4f" % (i, float(losses[i, 0])), flush=True)     # (a sign of life for whoever watches a long run)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    try:
        import psutil
        free, total = torch.cuda.mem_get_info()
        print("   after %d steps: process RSS %.0f MB, device memory in use %.0f MB" % (steps, psutil.Process().memory_info().rss / 1e6, (total - free) / 1e6), flush=True)
    except Exception:
        pass
    params = torch.cat([net.params.flatten(), net.mom.flatten(), net.bn.flatten()]).clone()       # variables, momentum slots, moving statistics
    return losses.cpu().numpy(), params.cpu().numpy(), dt


def main(B=16, steps=1500, lo=25, hi=85, net_type="dilated_grsl_rate8", channels=5, classes=6, comm_kind="none"):
    dev = "cuda:0"
    comm = None
    if comm_kind != "none":       # every collective of the step issued at world 1 (sums over one rank): library-side RCCL or the callback
        os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", DRS_FORCE_COLLECTIVES="1")
        os.environ["DRS_COMM"] = "rccl" if comm_kind == "rccl" else "torch"
        from drs_amd.dist import TorchComm
        torch.cuda.set_device(0)
        comm = TorchComm("nccl")
    tile, lab = make_tile(1024, 1024, channels, classes, seed=1234)
    pool = P.TilePool([tile], [lab], dev)
    l1, p1, t1 = run(B, steps, lo, hi, pool, dev, net_type, channels, classes, comm)
    l2, p2, t2 = run(B, steps, lo, hi, pool, dev, net_type, channels, classes, comm)
    if comm is not None:          # ... and they are identities: the run without them must give the same bits
        l3, p3, _ = run(B, steps, lo, hi, pool, dev, net_type, channels, classes, None)
        print("collectives=%s at world 1 against no collectives: identical %s" % (comm_kind, np.array_equal(l1, l3) and np.array_equal(p1, p3)))
    ok = np.isfinite(l1).all() and np.isfinite(p1).all()
    same = np.array_equal(l1, l2) and np.array_equal(p1, p2)
    print(net_type + " B=%d, %d steps, sides uniform in [%d, %d]: %.1f s and %.1f s; CE first 10 steps %.4f, last 10 steps %.4f; finite: %s; the two runs "
          "agree bit for bit (every loss, all %d variables): %s" % (B, steps, lo, hi, t1, t2, l1[:10, 0].mean(), l1[-10:, 0].mean(), ok, p1.size, same))
    if not (ok and same):
        bad = np.nonzero((l1 != l2).any(axis=1))[0]
        print("first differing step:", bad[:5])
        sys.exit(1)


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    main(int(kw.get("B", 16)), int(kw.get("steps", 1500)), int(kw.get("lo", 25)), int(kw.get("hi", 85)), kw.get("net", "dilated_grsl_rate8"),
         int(kw.get("channels", 5)), int(kw.get("classes", 6)), kw.get("comm", "none"))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
