#!/usr/bin/env python3
"""Every rotation angle 0..359 at several patch sides through the crop kernel against scipy.ndimage.rotate(order=0, reshape=False)
(what isprs:294-296 calls): source pixel, label and validity mask of every output pixel.  python tests/fuzz/check_rotation.py [S=9,15,16,25,33]"""
import os, sys
import numpy as np
import torch
from scipy import ndimage
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from drs_amd import patches as P
import test_gpu_patches as G


def main(sizes):
    nbad = 0
    for S in sizes:
        h = w = S + 6
        tile = (np.arange(h * w, dtype=np.float64) + 1).reshape(h, w, 1).repeat(3, axis=2)
        lab = (np.arange(h * w) % 6).reshape(h, w)
        pool = P.TilePool([tile], [lab], G.DEV, dtype=np.float64)
        B = 90
        net = G._net(3, B, S)
        for a0 in range(0, 360, B):
            inst = np.array([[0, 3, 2, a0 + k] for k in range(B)])
            aug = P.Augmentation(B)
            aug.rot_on[:] = 1
            for k in range(B):
                aug.rot[k] = P.rotation_params(a0 + k, S)
            P.crop_to_net(net, pool, inst, S, [0, 0, 0], [1, 1, 1], aug)
            torch.cuda.synchronize()
            a, Pd, ld = G._slab(net, B, S)
            got = a[:, Pd:Pd + S, Pd:Pd + S, 0]
            labs = net.labels[:B * S * S].cpu().numpy().reshape(B, S, S)
            mask = net.acc_mask[:B * S * S].cpu().numpy().reshape(B, S, S)
            for k in range(B):
                ang = a0 + k
                patch = tile[3:3 + S, 2:2 + S, 0]
                ref = ndimage.rotate(patch, ang, order=0, reshape=False)
                refl = ndimage.rotate(lab[3:3 + S, 2:2 + S], ang, order=0, reshape=False)
                refm = ndimage.rotate(np.ones((S, S), dtype=bool), ang, order=0, reshape=False)
                nd = int((got[k] != ref.astype(np.float32)).sum()) + int((labs[k] != refl).sum()) + int((mask[k].astype(bool) != refm).sum())
                if nd:
                    nbad += 1
                    print("S=%d angle %d: %d mismatches" % (S, ang, nd), flush=True)
    print("sides %s x 360 angles: %d (side, angle) pairs differ" % (sizes, nbad))
    sys.exit(1 if nbad else 0)


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    main([int(v) for v in kw.get("S", "9,15,16,25,33").split(",")])
