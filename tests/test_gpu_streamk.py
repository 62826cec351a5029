"""-m gpu: the stream-K form of the forward / input-gradient convolution (drs_conv_forward_ws, csrc/conv_mfma.hip).

Launches of fewer than 4096 output tiles -- every per-rank batch of a data-parallel run: 16 patches of 25..85 pixels a side
(BASELINE configs[2]; isprs:1727-1737 draws any integer size per step) -- cut the K-steps of all tiles into equal ranges, one per
workgroup; tiles that a cut crosses are completed from partial sums in a caller-owned workspace, in a fixed order.  Checked here:
every way a cut can fall (one workgroup for everything, cuts inside tiles, exactly on tile boundaries, more workgroups than
tiles, one K-step per workgroup) against the fp64 oracle, bias / accumulate / batch-norm statistics included; bitwise equality
with the plain kernel whenever no tile is cut; bitwise repeatability; and the per-rank sizes themselves under the library's own
rule against the oracle on sampled patches.
"""
import numpy as np
import pytest
import torch

from oracle import nets as onets
from oracle import tf_ops as T

pytestmark = pytest.mark.gpu

from gpu_util import DEV, conv_stats_moments, dev, padded, rel_err, stream   # noqa: E402


def _ws(lib, cout):
    n = lib.query("drs_conv_workspace_floats", cout)
    return torch.full((max(n, 1),), float("nan"), device=DEV), n          # poisoned: a piece read before it is written shows


@pytest.mark.parametrize("k,rate,cin,cout,B,S", [(3, 2, 64, 128, 3, 20), (3, 5, 128, 192, 2, 13), (5, 2, 64, 64, 2, 17), (4, 3, 64, 128, 1, 31),
                                                  (3, 8, 256, 256, 2, 12)])
def test_streamk_every_cut_matches_oracle(k, rate, cin, cout, B, S):
    from drs_amd import _lib
    lib = _lib.dev()            # libdrs_hip_dev.so: drs_debug_conv_splitk forces the number of workgroups
    rng = np.random.default_rng(k * 100 + rate * 10 + S)
    x = rng.normal(size=(B, S, S, cin)).astype(np.float32)
    w = (rng.normal(size=(k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
    bias = rng.normal(size=(cout,)).astype(np.float32)
    pb, pa = onets.same_pad(k, rate)
    P = max(pb, pa)
    M = B * S * S
    xd, wd, bd = padded(x, P), dev(w), dev(bias)
    ref = T.conv2d_same(x.astype(np.float64), w.astype(np.float64), rate) + bias.astype(np.float64)
    mt = lib.query("drs_conv_mtile", cout)
    bn = 192 if cout % 192 == 0 and cout % 128 else (128 if cout % 128 == 0 else 64)
    tiles = -(-M // 128) * (cout // bn)
    nks = k * k * cin // 32
    U = tiles * nks
    ws, nws = _ws(lib, cout)
    st = stream()
    plain = torch.empty(M, cout, device=DEV)
    pstats = torch.zeros(-(-M // mt) * cout * 2, device=DEV)
    lib.call("drs_conv_forward", xd.data_ptr(), B, S, P, cin, 0, wd.data_ptr(), bd.data_ptr(), k, rate, pb, cin, cout, plain.data_ptr(), cout, 0, 0,
             pstats.data_ptr(), st)
    # the PRODUCT library (libdrs_hip.so) under its own rule for the cut: held to the oracle itself, and the development library at
    # its default (-1 = the same rule) must give the same bits -- the forced cuts below then only vary the development binary
    pws, pnws = _ws(_lib, cout)
    pout = torch.full((M, cout + 32), -3.0, device=DEV)
    pst = torch.zeros(-(-M // mt) * cout * 2, device=DEV)
    _lib.call("drs_conv_forward_ws", xd.data_ptr(), B, S, P, cin, 0, wd.data_ptr(), bd.data_ptr(), k, rate, pb, cin, cout, pout.data_ptr(), cout + 32, 32,
              0, pst.data_ptr(), pws.data_ptr(), pnws, st)
    dout = torch.full((M, cout + 32), -3.0, device=DEV)
    dst = torch.zeros(-(-M // mt) * cout * 2, device=DEV)
    lib.drs_debug_conv_splitk(-1)
    lib.call("drs_conv_forward_ws", xd.data_ptr(), B, S, P, cin, 0, wd.data_ptr(), bd.data_ptr(), k, rate, pb, cin, cout, dout.data_ptr(), cout + 32, 32,
             0, dst.data_ptr(), ws.data_ptr(), nws, st)
    torch.cuda.synchronize()
    assert pnws == nws and mt == _lib.query("drs_conv_mtile", cout)
    got = pout.cpu().numpy()
    assert np.all(got[:, :32] == -3.0)
    assert rel_err(got[:, 32:].reshape(B, S, S, cout), ref) < 1e-5
    sv = conv_stats_moments(_lib, pst, M, mt, cout)
    assert np.abs(sv[:, 0] - ref.reshape(-1, cout).sum(0)).max() < 1e-5 * np.abs(ref.reshape(-1, cout)).sum(0).max()
    assert rel_err(sv[:, 1], (ref.reshape(-1, cout) ** 2).sum(0)) < 1e-5
    assert torch.equal(pout, dout) and torch.equal(pst, dst)
    whole = [W for W in range(1, tiles + 1) if U % W == 0 and (U // W) % nks == 0]
    cuts = sorted({1, 2, 3, 7, tiles - 1, tiles, tiles + 1, 2 * tiles, 3 * tiles + 1, U // 2, U - 1, U, U + 5, 768} | set(whole[:3]))
    try:
        for W in cuts:
            if W < 1:
                continue
            lib.drs_debug_conv_splitk(W)
            ws.fill_(float("nan"))
            out = torch.full((M, cout + 32), -3.0, device=DEV)
            stats = torch.zeros(-(-M // mt) * cout * 2, device=DEV)
            lib.call("drs_conv_forward_ws", xd.data_ptr(), B, S, P, cin, 0, wd.data_ptr(), bd.data_ptr(), k, rate, pb, cin, cout, out.data_ptr(), cout + 32,
                     32, 0, stats.data_ptr(), ws.data_ptr(), nws, st)
            torch.cuda.synchronize()
            got = out.cpu().numpy()
            assert np.all(got[:, :32] == -3.0), W
            assert rel_err(got[:, 32:].reshape(B, S, S, cout), ref) < 1e-5, W
            sv = conv_stats_moments(lib, stats, M, mt, cout)
            r2 = ref.reshape(-1, cout)
            assert np.abs(sv[:, 0] - r2.sum(0)).max() < 1e-5 * np.abs(r2).sum(0).max(), W
            assert rel_err(sv[:, 1], (r2 ** 2).sum(0)) < 1e-5, W
            if min(W, U) in whole:          # no tile is cut: the same K order per tile as the plain kernel, bit for bit
                assert torch.equal(out[:, 32:], plain) and torch.equal(stats, pstats), W
            # accumulate on top (the dense nets' input gradient, isprs:921-948), no bias
            lib.call("drs_conv_forward_ws", xd.data_ptr(), B, S, P, cin, 0, wd.data_ptr(), None, k, rate, pb, cin, cout, out.data_ptr(), cout + 32, 32, 1,
                     None, ws.data_ptr(), nws, st)
            torch.cuda.synchronize()
            assert rel_err(out.cpu().numpy()[:, 32:].reshape(B, S, S, cout), 2 * ref - bias.astype(np.float64)) < 1e-5, W
            # r05: the forms of the launch that change no sum -- whole tiles for the full rounds (hybrid) with the range first / the whole
            # tiles first / alternating, wave priority on or off -- agree bit for bit with each other; the r03 form that cuts every tile
            # (hybrid off) associates differently and is held to the oracle
            forms = {}
            for hyb, order, prio in ((1, 0, 0), (1, 1, 1), (1, 2, 1), (0, 0, 1)):
                lib.drs_debug_conv_hybrid(hyb); lib.drs_debug_conv_sk_order(order); lib.drs_debug_conv_prio(prio)
                ws.fill_(float("nan"))
                o3 = torch.full((M, cout + 32), -3.0, device=DEV)
                s3 = torch.zeros(-(-M // mt) * cout * 2, device=DEV)
                lib.call("drs_conv_forward_ws", xd.data_ptr(), B, S, P, cin, 0, wd.data_ptr(), bd.data_ptr(), k, rate, pb, cin, cout, o3.data_ptr(), cout + 32,
                         32, 0, s3.data_ptr(), ws.data_ptr(), nws, st)
                torch.cuda.synchronize()
                assert rel_err(o3.cpu().numpy()[:, 32:].reshape(B, S, S, cout), ref) < 1e-5, (W, hyb, order, prio)
                forms[(hyb, order, prio)] = (o3, s3)
            lib.drs_debug_conv_hybrid(1); lib.drs_debug_conv_sk_order(1); lib.drs_debug_conv_prio(-1)
            for key in ((1, 1, 1), (1, 2, 1)):
                assert torch.equal(forms[key][0], forms[(1, 0, 0)][0]) and torch.equal(forms[key][1], forms[(1, 0, 0)][1]), (W, key)
            # repeatable bit for bit
            out2 = torch.full((M, cout + 32), -3.0, device=DEV)
            lib.call("drs_conv_forward_ws", xd.data_ptr(), B, S, P, cin, 0, wd.data_ptr(), bd.data_ptr(), k, rate, pb, cin, cout, out2.data_ptr(), cout + 32,
                     32, 0, None, ws.data_ptr(), nws, st)
            lib.call("drs_conv_forward_ws", xd.data_ptr(), B, S, P, cin, 0, wd.data_ptr(), None, k, rate, pb, cin, cout, out2.data_ptr(), cout + 32, 32, 1,
                     None, ws.data_ptr(), nws, st)
            torch.cuda.synchronize()
            assert torch.equal(out, out2), W
        # a workspace too small for two pieces, or none: the plain kernel
        lib.drs_debug_conv_splitk(-1)
        out = torch.empty(M, cout, device=DEV)
        lib.call("drs_conv_forward_ws", xd.data_ptr(), B, S, P, cin, 0, wd.data_ptr(), bd.data_ptr(), k, rate, pb, cin, cout, out.data_ptr(), cout, 0, 0,
                 None, ws.data_ptr(), 100, st)
        torch.cuda.synchronize()
        assert torch.equal(out, plain)
        # switched off: the plain kernel too
        lib.drs_debug_conv_splitk(0)
        lib.call("drs_conv_forward_ws", xd.data_ptr(), B, S, P, cin, 0, wd.data_ptr(), bd.data_ptr(), k, rate, pb, cin, cout, out.data_ptr(), cout, 0, 0,
                 None, ws.data_ptr(), nws, st)
        torch.cuda.synchronize()
        assert torch.equal(out, plain)
    finally:
        lib.drs_debug_conv_splitk(-1)
        lib.drs_debug_conv_hybrid(1); lib.drs_debug_conv_sk_order(1); lib.drs_debug_conv_prio(-1)


def test_hybrid_geometry_keeps_every_workgroup_equal():
    """host side of the hybrid rule (sk_geometry): whole rounds of whole tiles + the remainder cut into one range per workgroup; no tile
    lost or taken twice, K-steps per workgroup within one of each other plus at most one whole tile's round-off"""
    import ctypes
    from drs_amd import _lib
    lib = _lib.dev()
    g3 = (ctypes.c_int * 3)()
    for bn in (64, 128, 192):
        for tiles in list(range(1, 1200, 7)) + [767, 768, 769, 1058, 1408, 1808, 2047, 3000, 4095]:
            for nks in (2, 16, 54, 72):
                G = lib.drs_debug_conv_sk_geometry(tiles, nks, bn, g3)
                if G == 0:
                    continue
                G, W, T = g3[0], g3[1], g3[2]
                assert 1 <= W <= G <= 768 and 1 <= T <= tiles and W <= T * nks
                assert W == min(G, T * nks)                          # a range for every workgroup (never an empty one)
                if T < tiles:
                    assert (tiles - T) % G == 0                      # whole rounds of whole tiles
                U = T * nks
                steps = [((w + 1) * U // W - w * U // W) for w in range(W)]
                assert sum(steps) == U and max(steps) - min(steps) <= 1


# conv8, conv3 (4x4, asymmetric padding), conv6 (one 192-wide tile), conv2 (64-wide tile) of Dilated8Pooling
RANK_SHAPES = [(3, 8, 256, 256), (4, 3, 64, 128), (3, 6, 192, 192), (5, 2, 64, 64)]


@pytest.mark.parametrize("B,S", [(16, 25), (16, 38), (16, 45), (16, 65), (16, 85), (32, 55)])
@pytest.mark.parametrize("k,rate,cin,cout", RANK_SHAPES)
def test_streamk_per_rank_sizes_match_oracle_on_sampled_patches(k, rate, cin, cout, B, S):
    """what one rank of an 8-GPU run of BASELINE configs[2] launches (batch 128 / 8, `uniform` over [25, 85]), under the product
    library's own rule for the cut: forward and input gradient against the fp64 oracle on sampled patches, the input gradient by
    adjointness with the forward on all of them, batch-norm statistics against the moments of the output."""
    from drs_amd import _lib
    M = B * S * S
    g0 = torch.Generator(device=DEV).manual_seed(100 * k + rate + S)
    pb, pa = onets.same_pad(k, rate)
    P = max(pb, pa)
    x = torch.randn(B, S, S, cin, device=DEV, generator=g0)
    g = torch.randn(B, S, S, cout, device=DEV, generator=g0)
    w = torch.randn(k, k, cin, cout, device=DEV, generator=g0) / (k * k * cin) ** 0.5
    bias = torch.randn(cout, device=DEV, generator=g0)
    xp = torch.nn.functional.pad(x, (0, 0, P, P, P, P)).contiguous()
    gp = torch.nn.functional.pad(g, (0, 0, P, P, P, P)).contiguous()
    st = stream()
    wt = torch.empty(w.numel(), device=DEV)
    _lib.call("drs_filter_flip_transpose", w.data_ptr(), wt.data_ptr(), k, cin, cout, st)
    nws = max(_lib.query("drs_conv_workspace_floats", cout), _lib.query("drs_conv_workspace_floats", cin))
    ws = torch.full((nws,), float("nan"), device=DEV)
    mt = _lib.query("drs_conv_mtile", cout)
    y = torch.empty(M, cout, device=DEV)
    stats = torch.zeros(-(-M // mt) * cout * 2, device=DEV)
    gx = torch.empty(M, cin, device=DEV)
    _lib.call("drs_conv_forward_ws", xp.data_ptr(), B, S, P, cin, 0, w.data_ptr(), bias.data_ptr(), k, rate, pb, cin, cout, y.data_ptr(), cout, 0, 0,
              stats.data_ptr(), ws.data_ptr(), nws, st)
    _lib.call("drs_conv_forward_ws", gp.data_ptr(), B, S, P, cout, 0, wt.data_ptr(), None, k, rate, pa, cout, cin, gx.data_ptr(), cin, 0, 0, None,
              ws.data_ptr(), nws, st)
    torch.cuda.synchronize()
    sample = sorted({0, B // 2, B - 1})
    w64, b64 = w.cpu().numpy().astype(np.float64), bias.cpu().numpy().astype(np.float64)
    xs, gs = x[sample].cpu().numpy().astype(np.float64), g[sample].cpu().numpy().astype(np.float64)
    ref = T.conv2d_same(xs, w64, rate) + b64
    gx_ref, _ = T.conv2d_same_bwd(xs, w64, rate, gs)
    assert rel_err(y.view(B, S, S, cout)[sample].cpu().numpy(), ref) < 1e-5
    assert rel_err(gx.view(B, S, S, cin)[sample].cpu().numpy(), gx_ref) < 1e-5
    yb = y.double() - bias.double()
    a = (yb * g.reshape(M, cout).double()).sum().item()
    b = (x.reshape(M, cin).double() * gx.double()).sum().item()
    assert abs(a - b) < 1e-6 * (yb.norm() * g.double().norm()).item(), (a, b)
    sv = conv_stats_moments(_lib, stats, M, mt, cout)
    y64 = y.double()
    mean, var = y64.mean(0).cpu().numpy(), y64.var(0, unbiased=False).cpu().numpy()
    np.testing.assert_allclose(sv[:, 0] / M, mean, rtol=0, atol=1e-6 * np.abs(mean).max() + 1e-7)
    np.testing.assert_allclose(sv[:, 1] / M - (sv[:, 0] / M) ** 2, var, rtol=2e-6)
    y2 = torch.empty(M, cout, device=DEV)
    _lib.call("drs_conv_forward_ws", xp.data_ptr(), B, S, P, cin, 0, w.data_ptr(), bias.data_ptr(), k, rate, pb, cin, cout, y2.data_ptr(), cout, 0, 0,
              None, ws.data_ptr(), nws, st)
    torch.cuda.synchronize()
    assert torch.equal(y, y2)
