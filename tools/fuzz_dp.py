#!/usr/bin/env python3
"""Randomised (net type, bands, classes, world size, per-rank batch, side) through the data-parallel training step: W processes
sharing the one GPU over gloo (the callback collectives), each with its shard, against the single-process step on the whole batch --
loss, gradients, updated variables, moving statistics, confusion matrix.  (The bounds are those of tests/test_gpu_dp.py: the sums
associate differently, so a few ReLU signs / pool winners flip.)      python tools/fuzz_dp.py [n=10] [seed=0]
With DRS_COMM=rccl FUZZ_DP_RCCL_LIB=<tests/c/nccl_shm_double.cpp built> in the environment the ranks' sums are issued by the LIBRARY over the
shared-memory stand-in for RCCL instead (the line says which path ran)."""
import os, sys, tempfile
import numpy as np
import torch
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def inputs(cfg):
    net, ch, K, W, b, S = cfg
    rng = np.random.default_rng(b * 100 + S)
    B = W * b
    return rng.normal(size=(B, S * S * ch)).astype(np.float32), rng.integers(0, K, size=(B, S * S))


def worker(rank, cfg, port, out):
    net, ch, K, W, b, S = cfg
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(W))
    if os.environ.get("FUZZ_DP_RCCL_LIB"):      # this tool's own variable: the library itself reads none (drs_rccl_bind_library is a call)
        from drs_amd import _lib
        _lib.call("drs_rccl_bind_library", os.environ["FUZZ_DP_RCCL_LIB"].encode())
    import torch.distributed as dist
    from drs_amd.dist import TorchComm, shard_slice
    from drs_amd.net import DilatedNet
    torch.cuda.set_device(0)
    comm = TorchComm("gloo")
    x, y = inputs(cfg)
    sl = shard_slice(W * b, rank, W)
    d = DilatedNet(net, ch, K, 0.005, b_max=b, s_max=S, device="cuda:0", seed=3, comm=comm)
    for _ in range(int(os.environ.get("FUZZ_DP_STEPS", "1"))):     # (more steps diverge by the NET's own sensitivity, tools/sensitivity.py: variables 3e-5 apart -> gradients 5e-2 apart)
        d.feed(x[sl], y[sl], S)
        res = d.train_step(b, S, 0.01)
    torch.cuda.synchronize()
    if rank == 0:
        np.savez(out, grads=d.grads.cpu().numpy(), params=d.params.cpu().numpy(), bn=d.bn.cpu().numpy(), loss=d.loss_value(res["loss_parts"]),
                 conf=res["conf"].cpu().numpy(), collectives=str(getattr(d, "collectives", "?")))
    comm.barrier()
    dist.destroy_process_group()


def main(n=10, seed=0):
    from drs_amd.net import DilatedNet
    from drs_amd.nets import known_net_types
    rng = np.random.default_rng(seed)
    nbad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for i in range(n):
            cfg = (str(rng.choice(known_net_types())), int(rng.choice([3, 4, 5])), int(rng.choice([2, 6])), int(rng.choice([2, 3, 4])),
                   int(rng.integers(1, 4)), int(rng.integers(9, 30)))
            if os.environ.get("FUZZ_DP_CFG"):
                c = os.environ["FUZZ_DP_CFG"].split(";")[i].split(",")
                cfg = (c[0], int(c[1]), int(c[2]), int(c[3]), int(c[4]), int(c[5]))
            out = os.path.join(tmp, "dp%d.npz" % i)
            mp.spawn(worker, args=(cfg, 29800 + (os.getpid() + i) % 150, out), nprocs=cfg[3], join=True)
            net, ch, K, W, b, S = cfg
            x, y = inputs(cfg)
            d = DilatedNet(net, ch, K, 0.005, b_max=W * b, s_max=S, device="cuda:0", seed=3)
            for _ in range(int(os.environ.get("FUZZ_DP_STEPS", "1"))):
                d.feed(x, y, S)
                res = d.train_step(W * b, S, 0.01)
            torch.cuda.synchronize()
            r = np.load(out)
            rel = lambda a, c: float(np.abs(a - c).max() / max(1e-30, np.abs(c).max()))
            e = dict(loss=abs(float(r["loss"]) - d.loss_value(res["loss_parts"])), grads=rel(r["grads"], d.grads.cpu().numpy()),
                     params=rel(r["params"], d.params.cpu().numpy()), bn=rel(r["bn"], d.bn.cpu().numpy()),
                     conf=int(np.abs(r["conf"].astype(np.int64) - res["conf"].cpu().numpy().astype(np.int64)).sum()))
            ok = e["loss"] < 1e-5 and e["grads"] < 1.5e-2 and e["params"] < 2e-4 and e["bn"] < 1e-6 and e["conf"] <= 2
            print("%s %s  loss %.1e grads %.1e params %.1e bn %.1e conf diff %d   [%s]" % ("ok  " if ok else "FAIL", cfg, e["loss"], e["grads"], e["params"], e["bn"], e["conf"],
                                                                                      str(r["collectives"])[:24]), flush=True)
            nbad += 0 if ok else 1
    print("%d data-parallel cases, %d failed" % (n, nbad))
    sys.exit(1 if nbad else 0)


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    main(int(kw.get("n", 10)), int(kw.get("seed", 0)))
