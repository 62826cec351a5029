#!/usr/bin/env python3
"""BASELINE.json configs beyond the headline line, on one GPU (the N-GPU forms shard the same loops over ranks):
  config 3: dilated_grsl_rate8 training, patch size drawn per step by `uniform` over [25, 85] (isprs:1727-1737), batch 128
  config 5: dilated_grsl_rate8 sliding-window inference of a 6000x6000x5 mosaic at 64x64 windows, stride 32 (isprs:1241-1284)
    python tools/bench_configs.py [arith=f32|bf16x3|bf16x6] [mosaic=6000] [steps=40]
    python tools/bench_configs.py which=124      configs 1, 2 and 4 (parity-test cases of BASELINE.json, timed here for the record):
  config 1: dilated_icpr_original (Dilated6) at 25x25x3, batch 16; config 2: dilated_grsl (Dilated6Pooling) at 64x64x5, batch 64;
  config 4: dilated_icpr_rate6_densely (DenseDilated6), `multinomial` over {25, 50, 75, 100}, 4 bands, 2 classes, batch 128
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drs_amd.net import DilatedNet  # noqa: E402
from drs_amd import loops, patches as P  # noqa: E402
from drs_amd.synthetic import make_tile, grid_instances  # noqa: E402

NET, CH, K = "dilated_grsl_rate8", 5, 6


def main(arith="f32", mosaic=6000, steps=40):
    dev = "cuda:0"
    # ---- config 3: variable patch size
    tile, lab = make_tile(2048, 2048, CH, K, seed=1234)
    pool = P.TilePool([tile], [lab], dev, dtype=np.float64)
    mean, std = tile[:, :, :3].mean(axis=(0, 1)).tolist(), tile[:, :, :3].std(axis=(0, 1)).tolist()
    B, values = 128, [25, 45, 65, 85]
    net = DilatedNet(NET, CH, K, 0.005, b_max=B, s_max=values[-1], device=dev, seed=42, arith=arith)
    inst = grid_instances(2048, 2048, values[-1], 25, B * 64, seed=0)
    np.random.seed(11)
    sizes = [P.draw_patch_size("uniform", values)[0] for _ in range(steps + 5)]

    def step(i):
        s = sizes[i]
        rows = inst[(i * B) % (len(inst) - B):(i * B) % (len(inst) - B) + B]
        aug = P.draw_augmentation(rows, s, CH, noise="device")
        P.crop_to_net(net, pool, rows, s, mean, std, aug)
        return net.train_step(B, s, 0.01)
    for i in range(5):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(5, steps + 5):
        step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    px = sum(B * s * s for s in sizes[5:])
    print("config 3 (%s): %d steps, sizes uniform[25,85] (mean %.1f): %.0f patches/s, %.1f Mpx/s trained, %.2f ms/step"
          % (arith, steps, np.mean(sizes[5:]), B * steps / dt, px / dt / 1e6, 1e3 * dt / steps), flush=True)
    del net, pool
    torch.cuda.empty_cache()
    # ---- config 5: whole-mosaic sliding window
    t0 = time.perf_counter()
    big, big_lab = make_tile(mosaic, mosaic, CH, K, seed=5, dtype=np.float32)
    tgen = time.perf_counter() - t0
    pool = P.TilePool([big], [big_lab], dev, dtype=np.float32)
    S, Bw = 64, 256
    net = DilatedNet(NET, CH, K, 0.005, b_max=Bw, s_max=S, device=dev, seed=42, arith=arith)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pred, _ = loops.predict_tile(net, pool, 0, S, Bw, mean, std)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nh, nw = P.window_counts(mosaic, mosaic, S, S // 2)
    print("config 5 (%s): %dx%d mosaic, %d windows of 64x64 at stride 32: %.2f s (%.1f M window-pixels/s, %.1f M map pixels/s); "
          "tile synthesis on the host %.1f s (not timed)" % (arith, mosaic, mosaic, nh * nw, dt, nh * nw * S * S / dt / 1e6,
                                                            mosaic * mosaic / dt / 1e6, tgen), flush=True)
    assert tuple(pred.shape) == (mosaic, mosaic)


def train_rate(net_type, ch, K, B, tile_side, draw, steps, label):
    dev = "cuda:0"
    tile, lab = make_tile(tile_side, tile_side, ch, K, seed=1234)
    pool = P.TilePool([tile], [lab], dev, dtype=np.float64)
    mean, std = tile[:, :, :3].mean(axis=(0, 1)).tolist(), tile[:, :, :3].std(axis=(0, 1)).tolist()
    np.random.seed(11)
    sizes = [draw() for _ in range(steps + 5)]
    s_max = max(sizes)
    net = DilatedNet(net_type, ch, K, 0.005, b_max=B, s_max=s_max, device=dev, seed=42)
    inst = grid_instances(tile_side, tile_side, s_max, 25, max(B * 64, 2 * B), seed=0)

    def step(i):
        s = sizes[i]
        o = (i * B) % max(1, len(inst) - B)
        rows = inst[o:o + B]
        aug = P.draw_augmentation(rows, s, ch, noise="device")
        P.crop_to_net(net, pool, rows, s, mean, std, aug)
        return net.train_step(B, s, 0.01)
    for i in range(5):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(5, steps + 5):
        step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    px = sum(B * s * s for s in sizes[5:])
    flops = 3 * 2 * net.plan.mac_per_pixel() * px
    print("%s: %s, batch %d, %d steps (mean side %.1f): %.0f patches/s, %.1f Mpx/s trained, %.2f ms/step, %.1f TFLOP/s (%.2f of the fp32 roof)"
          % (label, net_type, B, steps, np.mean(sizes[5:]), B * steps / dt, px / dt / 1e6, 1e3 * dt / steps, flops / dt / 1e12, flops / dt / 157.3e12),
          flush=True)
    del net, pool
    torch.cuda.empty_cache()


def main_124(steps):
    train_rate("dilated_icpr_original", 3, 6, 16, 256, lambda: 25, steps * 5, "config 1")
    train_rate("dilated_grsl", 5, 6, 64, 2048, lambda: 64, steps, "config 2")
    values = [25, 50, 75, 100]
    probs = P.define_multinomial_probs(values)
    train_rate("dilated_icpr_rate6_densely", 4, 2, 128, 500, lambda: P.draw_patch_size("multinomial", values, probs)[0], steps, "config 4")


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    if kw.get("which") == "124":
        main_124(int(kw.get("steps", 40)))
        sys.exit(0)
    main(kw.get("arith", "f32"), int(kw.get("mosaic", 6000)), int(kw.get("steps", 40)))
