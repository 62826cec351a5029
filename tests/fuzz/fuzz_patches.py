#!/usr/bin/env python3
"""Randomised tiles, instances and sizes through the crop + augment + normalise kernel (against oracle/host_ref.py's restatement of
dynamically_create_patches under the same numpy seed: bit-exact, rotations / flips / host noise included), the coffee / contest
samplers (flip by index, float16 patches and normalisation, void mask) and random tile / window / batch geometries through the
stitch kernels (tests/test_gpu_patches.py's check, called as a function).  Test infrastructure.
    python tests/fuzz/fuzz_patches.py [n=150] [seed=0]"""
import os, sys, traceback
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import host_ref as H
from drs_amd import patches as P
import test_gpu_patches as G


def crop_case(rng):
    C = int(rng.choice([3, 4, 5]))
    h, w = int(rng.integers(30, 140)), int(rng.integers(30, 140))
    S = int(rng.integers(5, min(h, w, 40) + 1))
    B = int(rng.integers(1, 9))
    nt = int(rng.integers(1, 3))
    tiles = [rng.uniform(size=(h, w, C)) for _ in range(nt)]
    labs = [rng.integers(0, 6, size=(h, w)) for _ in range(nt)]
    # instances anywhere, the border-clipped ones included (isprs:260-269 shifts them back); rotation angle in degrees
    inst = np.stack([rng.integers(0, nt, size=B), rng.integers(0, h - 1, size=B), rng.integers(0, w - 1, size=B), rng.integers(0, 360, size=B)], 1)
    mean = list(rng.uniform(0.2, 0.6, size=3)) + [0, 0]
    std = list(rng.uniform(0.1, 0.4, size=3)) + [1, 1]
    tag = "crop C%d %dx%d S%d B%d tiles%d" % (C, h, w, S, B, nt)
    net = G._net(C, B, S)
    pool = P.TilePool(tiles, labs, G.DEV, dtype=np.float64)
    bad = []
    for train in (False, True):
        seed = int(rng.integers(0, 2 ** 31))
        aug = None
        if train:
            np.random.seed(seed)
            aug = P.draw_augmentation(inst, S, C, noise="host")
        P.crop_to_net(net, pool, inst, S, mean[:C], std[:C], aug)
        torch.cuda.synchronize()
        np.random.seed(seed)
        want, wl, wm = H.dynamically_create_patches(tiles, labs, inst, S, is_train=train)
        H.normalize_images(want, mean, std)
        a, Pd, ld = G._slab(net, B, S)
        M = B * S * S
        if not np.array_equal(a[:, Pd:Pd + S, Pd:Pd + S, :C], want.astype(np.float32)):
            got = a[:, Pd:Pd + S, Pd:Pd + S, :C]
            for b in range(B):
                nd = int((got[b] != want[b].astype(np.float32)).any(axis=-1).sum())
                if nd:
                    bad.append("pixels train=%s: instance %d %s differs at %d of %d pixels, max |diff| %.3g; rot_on %s flip %s noise %s" % (
                        train, b, inst[b].tolist(), nd, S * S, float(np.abs(got[b] - want[b]).max()),
                        getattr(aug, "rot_on", [None] * B)[b] if aug else None, getattr(aug, "flip", [None] * B)[b] if aug else None,
                        getattr(aug, "noise_on", [None] * B)[b] if aug else None))
        halo = a.copy()
        halo[:, Pd:Pd + S, Pd:Pd + S] = 0
        if not (np.all(halo == 0) and np.all(a[:, :, :, C:] == 0)):
            bad.append("halo train=%s" % train)
        if not np.array_equal(net.labels[:M].cpu().numpy().reshape(-1, S, S), wl):
            bad.append("labels train=%s" % train)
        if not np.array_equal(net.acc_mask[:M].cpu().numpy().reshape(-1, S, S).astype(bool), np.asarray(wm).astype(bool)):
            bad.append("mask train=%s" % train)
    return tag, bad


def stitch_case(rng):
    S = int(rng.integers(4, 34))
    h, w = int(rng.integers(S, S + 90)), int(rng.integers(S, S + 90))
    K, bs = int(rng.integers(2, 8)), int(rng.integers(1, 70))
    tag = "stitch %dx%d S%d K%d batch %d" % (h, w, S, K, bs)
    try:
        G.test_stitch_matches_reference_order(h, w, S, K, bs)
        return tag, []
    except AssertionError:
        return tag, [traceback.format_exc().strip().splitlines()[-1][:160]]


def flavour_case(rng):
    """the coffee (flip by index, float16 patches, float16 in-place normalisation) and contest (void mask) samplers on the device
    against oracle/host_ref.py's restatement (itself held to the reference's functions by fuzz_vs_reference.py)"""
    coffee = bool(rng.integers(0, 2))
    nt = int(rng.integers(1, 3)) if coffee else 1
    h = int(rng.integers(24, 60))
    w = h if coffee else int(rng.integers(24, 60))
    K = 2 if coffee else 7
    data = rng.uniform(0, 1, size=(nt, h, w, 3)).astype(np.float32)
    lab = rng.integers(0, K + (0 if coffee else 1), size=(nt, h, w)).astype(np.uint8)
    S = int(rng.integers(5, min(h, w, 22) + 1))
    nd = int(rng.integers(1, 30))
    dist = [(int(rng.integers(0, nt)), int(rng.integers(0, h - 2)), int(rng.integers(0, w - 2))) for _ in range(nd)]
    B = int(rng.integers(1, 12))
    g = dict(shuffle=rng.integers(0, 3 * nd, size=B))
    tag = "%s %dx%d S%d windows %d B%d" % ("coffee" if coffee else "contest", h, w, S, nd, B)
    bad = []
    tiles, labs = [data[k] for k in range(nt)], [lab[k] for k in range(nt)]
    void = -1 if coffee else 7
    x, l, m = G._indexed_on_device(g, S, 3, K, dist, tiles, labs, void_label=void)
    p, c, q = H.indexed_create_patches(data, lab, S, dist, g["shuffle"], void_label=None if coffee else 7)
    if not np.array_equal(x, p.astype(np.float32)):
        bad.append("pixels")
    if not np.array_equal(l, c):
        bad.append("labels")
    if not coffee and not np.array_equal(m, q):
        bad.append("void mask")
    if coffee:
        p16, _, _ = H.indexed_create_patches(data, lab, S, dist, g["shuffle"], float16=True)
        xq, _, _ = G._indexed_on_device(g, S, 3, K, dist, tiles, labs, quantize=True)
        if not np.array_equal(xq, p16.astype(np.float32)):
            bad.append("float16 patches")
        mean, std = rng.uniform(0.2, 0.7, size=3).astype(np.float32), rng.uniform(0.1, 0.4, size=3).astype(np.float32)
        xn, _, _ = G._indexed_on_device(g, S, 3, K, dist, tiles, labs, quantize=True, mean=mean, std=std)
        if not np.array_equal(xn, H.normalize_images_f16(p16.copy(), mean, std).astype(np.float32)):
            bad.append("float16 normalisation")
    return tag, bad


def main(n=150, seed=0):
    rng = np.random.default_rng(seed)
    nbad = 0
    for i in range(n):
        tag, bad = (stitch_case if i % 4 == 0 else flavour_case if i % 4 == 1 else crop_case)(rng)
        if bad:
            nbad += 1
            print("FAIL", tag, bad, flush=True)
        elif i % 15 == 0:
            print("ok  ", tag, flush=True)
    print("%d cases, %d failed" % (n, nbad))
    sys.exit(1 if nbad else 0)


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    main(int(kw.get("n", 150)), int(kw.get("seed", 0)))
