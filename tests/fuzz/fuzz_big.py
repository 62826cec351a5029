#!/usr/bin/env python3
"""The convolution entry points at FULL batch sizes on random (layer shape of the nets, batch 16..128, side 25..100): forward and
input gradient against the fp64 oracle on sampled patches + adjointness + tile statistics (tests/test_gpu_streamk.py's check, called
as a function), and the filter gradient through the identity  sum(dW * W') == sum(conv(x, W') * g)  for a random W' (the forward
kernel, held to the oracle just before, supplies the right-hand side; fp64 reductions).  Test infrastructure.
    python tests/fuzz/fuzz_big.py [n=40] [seed=0]"""
import os, sys, traceback
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from drs_amd import _lib
from oracle import nets as onets
import test_gpu_streamk as K
from gpu_util import DEV, stream

SHAPES = [(5, 2, 64, 64), (4, 3, 64, 128), (4, 4, 128, 128), (3, 5, 128, 192), (3, 6, 192, 192), (3, 7, 192, 256), (3, 8, 256, 256),       # Dilated8Pooling
          (3, 5, 128, 256), (3, 6, 256, 256), (3, 4, 256, 256), (5, 1, 64, 64), (4, 2, 64, 128),                                        # Dilated6Pooling / Dilated6
          (5, 2, 32, 32), (4, 3, 64, 64), (4, 4, 128, 64), (3, 5, 192, 128), (3, 6, 320, 128)]                                          # DenseDilated6


def wgrad_identity(k, rate, cin, cout, B, S, seed):
    M = B * S * S
    g0 = torch.Generator(device=DEV).manual_seed(seed)
    pb, pa = onets.same_pad(k, rate)
    P = max(pb, pa)
    x = torch.randn(B, S, S, cin, device=DEV, generator=g0)
    g = torch.randn(B, S, S, cout, device=DEV, generator=g0)
    w2 = torch.randn(k, k, cin, cout, device=DEV, generator=g0) / (k * k * cin) ** 0.5
    xp = torch.nn.functional.pad(x, (0, 0, P, P, P, P)).contiguous()
    gp = torch.nn.functional.pad(g, (0, 0, P, P, P, P)).contiguous()
    st = stream()
    y = torch.empty(M, cout, device=DEV)
    _lib.call("drs_conv_forward", xp.data_ptr(), B, S, P, cin, 0, w2.data_ptr(), None, k, rate, pb, cin, cout, y.data_ptr(), cout, 0, 0, None, st)
    nsplit = _lib.query("drs_conv_wgrad_splits", B, S, k, cin, cout)
    slab = torch.full((nsplit * k * k * cin * cout,), float("nan"), device=DEV)
    gw = torch.full((k * k * cin * cout,), float("nan"), device=DEV)
    _lib.call("drs_conv_wgrad", xp.data_ptr(), B, S, P, cin, 0, gp.data_ptr(), P, cout, 0, k, rate, pb, cin, cin, cout, slab.data_ptr(), gw.data_ptr(), st)
    torch.cuda.synchronize()
    a = (y.double() * g.reshape(M, cout).double()).sum().item()
    b = (gw.double() * w2.reshape(-1).double()).sum().item()
    scale = (y.double().norm() * g.double().norm()).item()
    return abs(a - b) / scale, torch.isfinite(gw).all().item()


def main(n=40, seed=0, sides=""):
    """sides=32,64,96,128: draw the side from this list instead of 25..100 (the patch sides whose tiles are whole image rows take the
    full-tiles-first launch order with halo-tap skipping at every batch: r04)"""
    rng = np.random.default_rng(seed)
    nbad = 0
    side_list = [int(v) for v in sides.split(",")] if sides else None
    for i in range(n):
        k, rate, cin, cout = SHAPES[int(rng.integers(0, len(SHAPES)))]
        B = int(rng.choice([8, 16, 32, 64, 128])) if side_list else int(rng.choice([16, 32, 64, 128]))
        S = int(rng.choice(side_list)) if side_list else int(rng.integers(25, 101))
        while B * S * S * (cin + cout) * 4 * 3 > 6e9:            # keep the operands of one case within a few GB
            B //= 2
        args = (k, rate, cin, cout, B, S)
        bad = []
        try:
            K.test_streamk_per_rank_sizes_match_oracle_on_sampled_patches(k, rate, cin, cout, B, S)
        except AssertionError:
            bad.append("fwd/dgrad: " + traceback.format_exc().strip().splitlines()[-3].strip()[:160])
        e, finite = wgrad_identity(k, rate, cin, cout, B, S, int(rng.integers(0, 2 ** 31)))
        if not (e < 2e-6 and finite):
            bad.append("wgrad identity %.2e finite %s" % (e, finite))
        if bad:
            nbad += 1
            print("FAIL", args, bad, flush=True)
        elif i % 5 == 0:
            print("ok  ", args, "wgrad identity %.1e" % e, flush=True)
    print("%d cases, %d failed" % (n, nbad))
    sys.exit(1 if nbad else 0)


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    main(int(kw.get("n", 40)), int(kw.get("seed", 0)), kw.get("sides", ""))
