#!/usr/bin/env python3
"""The per-rank regime of BASELINE configs[2] on 8 GPUs (16 patches per rank, `uniform` over [25, 85], isprs:1727-1737): one
training step per (local batch, patch side), timed end to end and per kernel family (HIP events inside the step engine), with the
achieved fp32 TFLOP/s of the convolution families against the 157.3 TFLOP/s MFMA roof.

    python tools/bench_small_m.py [B=16,32] [S=25,35,45,55,65,75,85] [steps=20] [out=profiles/r03/small_m.json] [net=dilated_grsl_rate8]

Prints a table and writes the JSON; `weighted` = throughput of the size sequence a uniform draw over the listed sizes gives
(sum of patches / sum of step times)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drs_amd.net import DilatedNet  # noqa: E402
from drs_amd import patches as P  # noqa: E402
from drs_amd.synthetic import make_tile, grid_instances  # noqa: E402

PEAK = 157.3e12
CONV = ("conv_fwd", "conv_dgrad", "conv_wgrad")


def main(Bs, Ss, steps, out, net_type, channels, K):
    dev = "cuda:0"
    tile, lab = make_tile(1024, 1024, channels, K, seed=1234)
    pool = P.TilePool([tile], [lab], dev)
    net = DilatedNet(net_type, channels, K, 0.005, b_max=max(Bs), s_max=max(Ss), device=dev)
    rows = []
    for B in Bs:
        for S in Ss:
            inst = grid_instances(1024, 1024, S, 25, 4096, seed=0)
            np.random.seed(0)

            def step(i):
                r = inst[(i * B) % 4000:(i * B) % 4000 + B]
                aug = P.draw_augmentation(r, S, channels, noise="device")
                P.crop_to_net(net, pool, r, S, [0.5] * 3, [0.2] * 3, aug)
                return net.train_step(B, S, 0.01)
            for i in range(4):
                step(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                step(i)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            net.timer = True
            for i in range(5):
                step(i)
            summ = net.timer.summary()
            net.timer = None
            fam = {k: dict(ms=d["ms"] / 5, launches=d["launches"] // 5,
                           tflops=(d["work"] / (d["ms"] * 1e-3) / 1e12 if k in CONV else None)) for k, d in summ.items()}
            fd = [summ[k] for k in ("conv_fwd", "conv_dgrad") if k in summ]
            fd_tf = sum(d["work"] for d in fd) / (sum(d["ms"] for d in fd) * 1e-3) / 1e12
            wg_tf = fam["conv_wgrad"]["tflops"]
            ksum = sum(d["ms"] for d in fam.values())
            rows.append(dict(B=B, S=S, ms=dt * 1e3, patches_per_s=B / dt, kernels_ms=ksum, fwd_dgrad_tflops=fd_tf, fwd_dgrad_frac=fd_tf * 1e12 / PEAK,
                             wgrad_tflops=wg_tf, wgrad_frac=wg_tf * 1e12 / PEAK, families=fam))
            print("B=%3d S=%3d  %7.3f ms/step %7.0f patches/s | kernels %6.3f ms | fwd+dgrad %6.1f TF (%.3f)  wgrad %6.1f TF (%.3f) | %s" % (
                B, S, dt * 1e3, B / dt, ksum, fd_tf, fd_tf * 1e12 / PEAK, wg_tf, wg_tf * 1e12 / PEAK,
                "  ".join("%s %.3f" % (k[:12], v["ms"]) for k, v in sorted(fam.items()))), flush=True)
    res = dict(net=net_type, device=torch.cuda.get_device_name(0), rows=rows, weighted={})
    for B in Bs:
        rs = [r for r in rows if r["B"] == B]
        res["weighted"][str(B)] = dict(patches_per_s=B * len(rs) / sum(r["ms"] * 1e-3 for r in rs),
                                       px_per_s=sum(B * r["S"] ** 2 for r in rs) / sum(r["ms"] * 1e-3 for r in rs))
        print("B=%d size-weighted: %.0f patches/s per rank, %.2f Mpx/s" % (B, res["weighted"][str(B)]["patches_per_s"], res["weighted"][str(B)]["px_per_s"] / 1e6))
    if out:
        os.makedirs(os.path.dirname(out), exist_ok=True)
        with open(out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    main([int(v) for v in kw.get("B", "16,32").split(",")], [int(v) for v in kw.get("S", "25,35,45,55,65,75,85").split(",")],
         int(kw.get("steps", 20)), kw.get("out", "profiles/r03/small_m.json"), kw.get("net", "dilated_grsl_rate8"),
         int(kw.get("channels", 5)), int(kw.get("K", 6)))
