#!/usr/bin/env python3
"""Randomised shapes through the convolution entry points against the fp64 oracle: forward (plain and with the stream-K workspace),
input gradient, filter gradient -- kernel 1..5, rate 1..9, channels in steps of 32 / 64, batches and sides that leave ragged tiles and
partial chunks, halos wider than needed, slices of wider slabs, NaN-poisoned scratch.  Test infrastructure (uses oracle/); prints
every case that misses 1e-5.      python tests/fuzz/fuzz_ops.py [n=200] [seed=0]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from drs_amd import _lib
from oracle import nets as onets
from oracle import tf_ops as T
from gpu_util import DEV, conv_stats_moments, dev, padded, rel_err, stream


def case(rng, lib):
    k = int(rng.integers(1, 6))
    rate = int(rng.integers(1, 10)) if k > 1 else 1
    cin = int(rng.choice([32, 64, 96, 128, 192, 256]))
    cout = int(rng.choice([32, 64, 128, 192, 256]))
    S = int(rng.integers(3, 40))
    B = int(rng.integers(1, 7))
    while B * S * S * k * k * cin * cout > 3e10:      # keep the numpy oracle to a second or so
        S = max(3, S - 5)
    extra = int(rng.integers(0, 3))
    coff = int(rng.choice([0, 32]))
    x = rng.normal(size=(B, S, S, cin)).astype(np.float32)
    w = (rng.normal(size=(k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
    bias = rng.normal(size=(cout,)).astype(np.float32)
    g = rng.normal(size=(B, S, S, cout)).astype(np.float32)
    pb, pa = onets.same_pad(k, rate)
    P = max(pb, pa) + extra
    M = B * S * S
    tag = "k%d r%d %3d->%3d B%d S%2d P%d coff%d" % (k, rate, cin, cout, B, S, P, coff)
    xd = padded(x, P, ld=cin + coff, coff=coff, fill=7.0)
    wd, bd = dev(w), dev(bias)
    mt = lib.query("drs_conv_mtile", cout)
    rows = (M + mt - 1) // mt
    ref = T.conv2d_same(x.astype(np.float64), w.astype(np.float64), rate) + bias.astype(np.float64)
    gx_ref, gw_ref = T.conv2d_same_bwd(x.astype(np.float64), w.astype(np.float64), rate, g.astype(np.float64))
    bad = []
    nws = max(lib.query("drs_conv_workspace_floats", cout), lib.query("drs_conv_workspace_floats", cin))
    ws = torch.full((max(nws, 4),), float("nan"), device=DEV)
    for form in ("plain", "ws"):
        out = torch.full((M, cout + 32), -3.0, dtype=torch.float32, device=DEV)
        stats = torch.full((rows * cout * 2,), float("nan"), dtype=torch.float32, device=DEV)
        if form == "plain":
            lib.call("drs_conv_forward", xd.data_ptr(), B, S, P, cin + coff, coff, wd.data_ptr(), bd.data_ptr(), k, rate, pb, cin, cout,
                     out.data_ptr(), cout + 32, 32, 0, stats.data_ptr(), stream())
        else:
            lib.call("drs_conv_forward_ws", xd.data_ptr(), B, S, P, cin + coff, coff, wd.data_ptr(), bd.data_ptr(), k, rate, pb, cin, cout,
                     out.data_ptr(), cout + 32, 32, 0, stats.data_ptr(), ws.data_ptr(), nws, stream())
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        e = rel_err(got[:, 32:].reshape(B, S, S, cout), ref)
        if not e < 1e-5 or not np.all(got[:, :32] == -3.0):
            bad.append("fwd/%s %.2e" % (form, e))
        st = conv_stats_moments(lib, stats, M, mt, cout)
        r2 = ref.reshape(-1, cout)
        if not (np.abs(st[:, 0] - r2.sum(axis=0)).max() < 1e-5 * np.abs(r2).sum(axis=0).max() and rel_err(st[:, 1], (r2 ** 2).sum(axis=0)) < 1e-5):
            bad.append("stats/%s" % form)
    gd = padded(g, P, ld=cout, coff=0)
    wt = torch.zeros(k * k * cin * cout, dtype=torch.float32, device=DEV)
    lib.call("drs_filter_flip_transpose", wd.data_ptr(), wt.data_ptr(), k, cin, cout, stream())
    for form in ("plain", "ws"):
        gx = torch.full((M * cin,), float("nan"), dtype=torch.float32, device=DEV)
        if form == "plain":
            lib.call("drs_conv_forward", gd.data_ptr(), B, S, P, cout, 0, wt.data_ptr(), None, k, rate, pa, cout, cin, gx.data_ptr(), cin, 0, 0, None, stream())
        else:
            ws.fill_(float("nan"))
            lib.call("drs_conv_forward_ws", gd.data_ptr(), B, S, P, cout, 0, wt.data_ptr(), None, k, rate, pa, cout, cin, gx.data_ptr(), cin, 0, 0, None,
                     ws.data_ptr(), nws, stream())
        torch.cuda.synchronize()
        e = rel_err(gx.cpu().numpy().reshape(B, S, S, cin), gx_ref)
        if not e < 1e-5:
            bad.append("dgrad/%s %.2e" % (form, e))
    nsplit = lib.query("drs_conv_wgrad_splits", B, S, k, cin, cout)
    slab = torch.full((nsplit * k * k * cin * cout,), float("nan"), dtype=torch.float32, device=DEV)
    gw = torch.full((k * k * cin * cout,), float("nan"), dtype=torch.float32, device=DEV)
    lib.call("drs_conv_wgrad", xd.data_ptr(), B, S, P, cin + coff, coff, gd.data_ptr(), P, cout, 0, k, rate, pb, cin, cin, cout,
             slab.data_ptr(), gw.data_ptr(), stream())
    torch.cuda.synchronize()
    e = rel_err(gw.cpu().numpy().reshape(k, k, cin, cout), gw_ref)
    if not e < 1e-5:
        bad.append("wgrad %.2e" % e)
    return tag, bad


def main(n=200, seed=0):
    lib = _lib
    rng = np.random.default_rng(seed)
    nbad = 0
    for i in range(n):
        tag, bad = case(rng, lib)
        if bad:
            nbad += 1
            print("FAIL", tag, bad, flush=True)
        elif i % 20 == 0:
            print("ok  ", tag, flush=True)
    print("%d cases, %d failed" % (n, nbad))
    sys.exit(1 if nbad else 0)


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    main(int(kw.get("n", 200)), int(kw.get("seed", 0)))
