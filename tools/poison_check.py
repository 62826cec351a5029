#!/usr/bin/env python3
"""Does a training step read any scratch it has not written?  The same step twice from the same state: once with the library's
scratch buffers as they are, once with every one of them filled with NaN (floats) / 0xFF (bytes, ints) first.  State that must
persist (variables, momentum, moving statistics, the activation slabs' zero halos, labels / masks / the input slab) is left alone;
everything else is fair game.  Results must agree bit for bit.   python tools/poison_check.py [B=16] [S=37] [net=...]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drs_amd.net import DilatedNet
from drs_amd import patches as P
from drs_amd.synthetic import make_tile, grid_instances

KEEP = ("params", "momentum", "bn", "labels", "acc_mask", "loss_mask", "w0pad")


def one(B, S, net_type, channels, classes, poison, pool, dev, steps=3):
    net = DilatedNet(net_type, channels, classes, 0.005, b_max=B, s_max=S, device=dev, seed=42)
    inst = grid_instances(512, 512, S, 25, 512, seed=3)
    np.random.seed(5)
    outs = []
    skipped = []
    for i in range(steps):
        rows = inst[i * B:(i + 1) * B]
        aug = P.draw_augmentation(rows, S, channels, noise="device")
        if poison:
            for name, t in net._bufs.items():
                if name in KEEP or name.startswith("act:"):
                    skipped.append(name)
                    continue
                if t.dtype in (torch.float32, torch.float64):
                    t.fill_(float("nan"))
                else:
                    t.fill_(-1 if t.dtype in (torch.int32, torch.int64) else 255)
        P.crop_to_net(net, pool, rows, S, [0.5] * 3, [0.2] * 3, aug)
        o = net.train_step(B, S, 0.01)
        torch.cuda.synchronize()
        outs.append((o["loss_parts"].clone().cpu().numpy(), o["pred"].clone().cpu().numpy(), o["conf"].clone().cpu().numpy()))
    state = torch.cat([net.params.flatten(), net.mom.flatten(), net.bn.flatten()]).cpu().numpy()
    return outs, state, sorted(set(skipped)), sorted(net._bufs)


def main(B=16, S=37, net_type="dilated_grsl_rate8", channels=5, classes=6):
    dev = "cuda:0"
    tile, lab = make_tile(512, 512, channels, classes, seed=1234)
    pool = P.TilePool([tile], [lab], dev)
    a, sa, _, names = one(B, S, net_type, channels, classes, False, pool, dev)
    b, sb, kept, _ = one(B, S, net_type, channels, classes, True, pool, dev)
    ok = all(np.array_equal(x[j], y[j]) for x, y in zip(a, b) for j in range(3)) and np.array_equal(sa, sb) and np.isfinite(sa).all()
    print("%s B=%d S=%d: %d buffers, %d left alone (%s ...); poisoned scratch gives the same bits: %s" % (net_type, B, S, len(names), len(kept), ", ".join(kept[:6]), ok))
    if not ok:
        for i, (x, y) in enumerate(zip(a, b)):
            print(" step", i, "loss", x[0], y[0], "pred equal", np.array_equal(x[1], y[1]), "conf equal", np.array_equal(x[2], y[2]))
        sys.exit(1)


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    main(int(kw.get("B", 16)), int(kw.get("S", 37)), kw.get("net", "dilated_grsl_rate8"), int(kw.get("channels", 5)), int(kw.get("classes", 6)))
