#!/usr/bin/env python3
"""Schedule-fuzz soak (development library): a mixed-size training loop at the per-rank batch, with and without every collective of the
step forced on at world 1 (each form of the library-side collectives), run under several seeds of sleeps on the step's streams
(drs_debug_jitter: up to 150 us, one time in eight up to 4 ms, at every cross-stream hand-over) -- every loss and every variable
must be the unjittered, collective-free run's bit for bit.    python tools/jitter_soak.py [B=16] [steps=150] [seeds=4] [net=dilated_grsl_rate8 channels=5 classes=6]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drs_amd import _lib
d = _lib.dev()
_lib._lib = d.lib
from drs_amd.net import DilatedNet
from drs_amd import patches as P
from drs_amd.synthetic import make_tile, grid_instances

DEV = "cuda:0"


def run(B, steps, sizes, pool, comm, net_type="dilated_grsl_rate8", ch=5, K=6):
    net = DilatedNet(net_type, ch, K, 0.005, b_max=B, s_max=max(sizes), device=DEV, seed=42, comm=comm)
    rng = np.random.default_rng(7)
    np.random.seed(11)
    inst = {S: grid_instances(512, 512, S, 25, 1024, seed=S) for S in sizes}
    losses = torch.zeros(steps, 2, dtype=torch.float64, device=DEV)
    for i in range(steps):
        S = int(sizes[rng.integers(0, len(sizes))])
        rows = inst[S][(i * B) % 900:(i * B) % 900 + B]
        aug = P.draw_augmentation(rows, S, ch, noise="device")
        P.crop_to_net(net, pool, rows, S, [0.5] * 3, [0.2] * 3, aug)
        losses[i] = net.train_step(B, S, 0.01)["loss_parts"]
    torch.cuda.synchronize()
    state = torch.cat([net.params.flatten(), net.mom.flatten(), net.bn.flatten()]).cpu().numpy()
    label = getattr(net, "collectives", "none")
    net.close()
    return losses.cpu().numpy(), state, label


def main(B=16, steps=150, nseeds=4, net_type="dilated_grsl_rate8", ch=5, K=6):
    os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29537", DRS_FORCE_COLLECTIVES="1", DRS_COMM="rccl")
    from drs_amd.dist import TorchComm
    torch.cuda.set_device(0)
    comm = TorchComm("nccl")
    tile, lab = make_tile(512, 512, ch, K, seed=1234)
    pool = P.TilePool([tile], [lab], DEV)
    sizes = (25, 33, 38, 45, 55, 61, 64, 75)
    d.drs_debug_jitter(0)
    ref_l, ref_s, _ = run(B, steps, sizes, pool, None, net_type, ch, K)
    bad = 0
    for form, env in (("none", None), ("inline", {}), ("buckets", {"DRS_RCCL_BUCKETS": "2"}), ("async", {"DRS_RCCL_ASYNC": "1"})):
        for k in ("DRS_RCCL_ASYNC", "DRS_RCCL_BUCKETS"):
            os.environ.pop(k, None)
        if env:
            os.environ.update(env)
        for j in range(nseeds):
            seed = 0 if j == 0 else 1000003 * j + 17
            d.drs_debug_jitter(seed)
            l, s, label = run(B, steps, sizes, pool, comm if env is not None else None, net_type, ch, K)
            same = np.array_equal(l, ref_l) and np.array_equal(s, ref_s)
            bad += 0 if same else 1
            print("%-8s [%s] jitter seed %-10d: %d steps, identical to the unjittered collective-free run: %s" % (form, str(label)[:28], seed, steps, same), flush=True)
    d.drs_debug_jitter(0)
    print("%d runs differ" % bad)
    torch.distributed.destroy_process_group()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    main(int(kw.get("B", 16)), int(kw.get("steps", 150)), int(kw.get("seeds", 4)), kw.get("net", "dilated_grsl_rate8"), int(kw.get("channels", 5)), int(kw.get("classes", 6)))
