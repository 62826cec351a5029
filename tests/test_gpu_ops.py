"""-m gpu: every HIP entry point of include/drs.h against the CPU oracle (oracle/tf_ops.py, fp64) on the
same seeded inputs.  Tolerances: fp32 kernels vs fp64 oracle, relative to the tensor's max magnitude,
1e-5 for single ops (BASELINE north_star: logits within 1e-3 relative end to end); integer outputs exact."""
import numpy as np
import pytest
import torch

from oracle import tf_ops as T
from oracle import nets as onets

pytestmark = pytest.mark.gpu

from gpu_util import DEV, conv_stats_moments, dev, padded, rel_err, stream, unpad   # noqa: E402


@pytest.fixture(scope="module")
def lib():
    from drs_amd import _lib
    assert torch.cuda.is_available()
    _lib.load()
    return _lib


# every (k, rate) of the four nets with channel counts covering each tile configuration of the kernels
CONV_CASES = [
    (5, 1, 32, 64, 3, 9), (5, 2, 64, 64, 2, 12), (4, 3, 64, 128, 2, 11), (4, 4, 128, 128, 2, 10),
    (3, 5, 128, 192, 2, 13), (3, 6, 192, 192, 1, 15), (3, 7, 192, 256, 1, 16), (3, 8, 256, 256, 2, 17),
    (5, 1, 32, 32, 2, 25), (4, 2, 64, 128, 2, 8), (3, 4, 128, 256, 1, 9), (3, 6, 320, 128, 1, 14),
    (4, 4, 128, 64, 1, 12), (3, 5, 192, 128, 1, 12), (5, 2, 32, 32, 3, 7),
]


@pytest.mark.parametrize("k,rate,cin,cout,B,S", CONV_CASES)
def test_conv_forward_dgrad_wgrad(lib, k, rate, cin, cout, B, S):
    rng = np.random.default_rng(k * 1000 + rate * 100 + cin + cout + S)
    x = rng.normal(size=(B, S, S, cin)).astype(np.float32)
    w = (rng.normal(size=(k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
    bias = rng.normal(size=(cout,)).astype(np.float32)
    g = rng.normal(size=(B, S, S, cout)).astype(np.float32)
    pb, pa = onets.same_pad(k, rate)
    P = max(pb, pa) + 1                      # a halo wider than needed must also work
    M = B * S * S
    xd = padded(x, P, ld=cin + 32, coff=32, fill=7.0)      # slice of a wider slab; foreign channels are junk
    wd, bd = dev(w), dev(bias)
    out = torch.full((M, cout + 32), -3.0, dtype=torch.float32, device=DEV)
    mt = lib.query("drs_conv_mtile", cout)
    rows = (M + mt - 1) // mt
    stats = torch.zeros(rows * cout * 2, dtype=torch.float32, device=DEV)
    lib.call("drs_conv_forward", xd.data_ptr(), B, S, P, cin + 32, 32, wd.data_ptr(), bd.data_ptr(), k, rate, pb, cin, cout,
             out.data_ptr(), cout + 32, 32, 0, stats.data_ptr(), stream())
    torch.cuda.synchronize()
    ref = T.conv2d_same(x.astype(np.float64), w.astype(np.float64), rate) + bias.astype(np.float64)
    got = out.cpu().numpy()
    assert rel_err(got[:, 32:].reshape(B, S, S, cout), ref) < 1e-5
    assert np.all(got[:, :32] == -3.0)                      # neighbouring channels untouched
    st = conv_stats_moments(lib, stats, M, mt, cout)
    r2 = ref.reshape(-1, cout)
    assert np.abs(st[:, 0] - r2.sum(axis=0)).max() < 1e-5 * np.abs(r2).sum(axis=0).max()
    assert rel_err(st[:, 1], (r2 ** 2).sum(axis=0)) < 1e-5
    # accumulate mode
    lib.call("drs_conv_forward", xd.data_ptr(), B, S, P, cin + 32, 32, wd.data_ptr(), None, k, rate, pb, cin, cout,
             out.data_ptr(), cout + 32, 32, 1, None, stream())
    torch.cuda.synchronize()
    ref2 = ref + T.conv2d_same(x.astype(np.float64), w.astype(np.float64), rate)
    assert rel_err(out.cpu().numpy()[:, 32:].reshape(B, S, S, cout), ref2) < 1e-5

    # gradients
    gx_ref, gw_ref = T.conv2d_same_bwd(x.astype(np.float64), w.astype(np.float64), rate, g.astype(np.float64))
    gd = padded(g, P, ld=cout, coff=0)
    wt = torch.zeros(k * k * cin * cout, dtype=torch.float32, device=DEV)
    lib.call("drs_filter_flip_transpose", wd.data_ptr(), wt.data_ptr(), k, cin, cout, stream())
    gx = torch.zeros(M * cin, dtype=torch.float32, device=DEV)
    lib.call("drs_conv_forward", gd.data_ptr(), B, S, P, cout, 0, wt.data_ptr(), None, k, rate, pa, cout, cin, gx.data_ptr(),
             cin, 0, 0, None, stream())
    nsplit = lib.query("drs_conv_wgrad_splits", B, S, k, cin, cout)
    slab = torch.zeros(nsplit * k * k * cin * cout, dtype=torch.float32, device=DEV)
    gw = torch.zeros(k * k * cin * cout, dtype=torch.float32, device=DEV)
    lib.call("drs_conv_wgrad", xd.data_ptr(), B, S, P, cin + 32, 32, gd.data_ptr(), P, cout, 0, k, rate, pb, cin, cin, cout,
             slab.data_ptr(), gw.data_ptr(), stream())
    torch.cuda.synchronize()
    assert rel_err(gx.cpu().numpy().reshape(B, S, S, cin), gx_ref) < 1e-5
    assert rel_err(gw.cpu().numpy().reshape(k, k, cin, cout), gw_ref) < 1e-5


@pytest.mark.parametrize("k,rate,cin,cout,B,S", [(3, 8, 64, 128, 3, 20), (3, 7, 64, 64, 2, 33), (4, 3, 64, 128, 5, 9), (5, 2, 32, 64, 2, 16),
                                                  (3, 5, 128, 192, 2, 27)])
def test_halo_tap_skipping_is_bitwise_neutral(lib, k, rate, cin, cout, B, S):
    """Filter taps / pixel chunks that meet only the zero halo are not executed on large grids (drs_common.hpp); forcing the
    skip on (2) and off (0) at small sizes must give bitwise identical forward, input-gradient and filter-gradient results, in
    the exact-fp32 and in the split-bf16 kernels (tiles crossing image boundaries, ragged chunks, odd S included)."""
    lib = lib.dev()          # libdrs_hip_dev.so: the same sources + the A/B switches of include/drs_dev.h
    rng = np.random.default_rng(S * 7 + cout)
    x = rng.normal(size=(B, S, S, cin)).astype(np.float32)
    w = (rng.normal(size=(k, k, cin, cout)) * 0.05).astype(np.float32)
    g = rng.normal(size=(B, S, S, cout)).astype(np.float32)
    pb, pa = onets.same_pad(k, rate)
    P = max(pb, pa)
    M = B * S * S
    xd, gd, wd = padded(x, P), padded(g, P), dev(w)
    wt = torch.zeros(w.size, dtype=torch.float32, device=DEV)
    lib.call("drs_filter_flip_transpose", wd.data_ptr(), wt.data_ptr(), k, cin, cout, stream())
    ns = 2
    xt = torch.zeros(ns * xd.numel(), dtype=torch.int16, device=DEV)
    gt = torch.zeros(ns * gd.numel(), dtype=torch.int16, device=DEV)
    lib.call("drs_split_terms", xd.data_ptr(), xd.numel(), ns, xt.data_ptr(), stream())
    lib.call("drs_split_terms", gd.data_ptr(), gd.numel(), ns, gt.data_ptr(), stream())
    wf = torch.zeros(ns * w.size, dtype=torch.int16, device=DEV)
    wg = torch.zeros(ns * w.size, dtype=torch.int16, device=DEV)
    lib.call("drs_filter_split", wd.data_ptr(), k, cin, cin, cout, ns, wf.data_ptr(), wg.data_ptr() if cin % 64 == 0 else None, stream())
    res = {}
    raw = lib
    try:
        for mode in (0, 2):
            raw.drs_debug_skip_taps(mode)
            out = torch.zeros(M * cout, dtype=torch.float32, device=DEV)
            gx = torch.zeros(M * cin, dtype=torch.float32, device=DEV)
            gw = torch.zeros(w.size, dtype=torch.float32, device=DEV)
            lib.call("drs_conv_forward", xd.data_ptr(), B, S, P, cin, 0, wd.data_ptr(), None, k, rate, pb, cin, cout, out.data_ptr(), cout,
                     0, 0, None, stream())
            lib.call("drs_conv_forward", gd.data_ptr(), B, S, P, cout, 0, wt.data_ptr(), None, k, rate, pa, cout, cin, gx.data_ptr(), cin,
                     0, 0, None, stream())
            nsp = lib.query("drs_conv_wgrad_splits", B, S, k, cin, cout)
            slab = torch.zeros(nsp * w.size, dtype=torch.float32, device=DEV)
            lib.call("drs_conv_wgrad", xd.data_ptr(), B, S, P, cin, 0, gd.data_ptr(), P, cout, 0, k, rate, pb, cin, cin, cout,
                     slab.data_ptr(), gw.data_ptr(), stream())
            out2 = torch.zeros(M * cout, dtype=torch.float32, device=DEV)
            gw2 = torch.zeros(w.size, dtype=torch.float32, device=DEV)
            gx2 = torch.zeros(M * cin, dtype=torch.float32, device=DEV)
            if cout % 64 == 0:
                lib.call("drs_conv_forward_split", xt.data_ptr(), B, S, P, cin, 0, wf.data_ptr(), None, k, rate, pb, cin, cout,
                         out2.data_ptr(), cout, 0, 0, None, ns, stream())
                nsp2 = lib.query("drs_conv_wgrad_split_splits", B, S, k, cin, cout, P, ns)
                slab2 = torch.zeros(nsp2 * w.size, dtype=torch.float32, device=DEV)
                lib.call("drs_conv_wgrad_split", xt.data_ptr(), B, S, P, cin, 0, gt.data_ptr(), P, cout, 0, k, rate, pb, cin, cin, cout,
                         slab2.data_ptr(), gw2.data_ptr(), ns, stream())
                if cin % 64 == 0:
                    lib.call("drs_conv_forward_split", gt.data_ptr(), B, S, P, cout, 0, wg.data_ptr(), None, k, rate, pa, cout, cin,
                             gx2.data_ptr(), cin, 0, 0, None, ns, stream())
            torch.cuda.synchronize()
            res[mode] = [t.cpu().numpy() for t in (out, gx, gw, out2, gx2, gw2)]
    finally:
        raw.drs_debug_skip_taps(1)
    for a, b, name in zip(res[0], res[2], ("fwd", "dgrad", "wgrad", "fwd split", "dgrad split", "wgrad split")):
        np.testing.assert_array_equal(a, b, err_msg=name)
    ref = T.conv2d_same(x.astype(np.float64), w.astype(np.float64), rate)
    assert rel_err(res[2][0].reshape(B, S, S, cout), ref) < 1e-5
    # the product library (libdrs_hip.so, its own rule: no skip at these sizes) gives the same bits as both forced forms
    from drs_amd import _lib as prod
    out = torch.zeros(M * cout, dtype=torch.float32, device=DEV)
    gx = torch.zeros(M * cin, dtype=torch.float32, device=DEV)
    gw = torch.zeros(w.size, dtype=torch.float32, device=DEV)
    prod.call("drs_conv_forward", xd.data_ptr(), B, S, P, cin, 0, wd.data_ptr(), None, k, rate, pb, cin, cout, out.data_ptr(), cout, 0, 0, None, stream())
    prod.call("drs_conv_forward", gd.data_ptr(), B, S, P, cout, 0, wt.data_ptr(), None, k, rate, pa, cout, cin, gx.data_ptr(), cin, 0, 0, None, stream())
    nsp = prod.query("drs_conv_wgrad_splits", B, S, k, cin, cout)
    slab = torch.zeros(nsp * w.size, dtype=torch.float32, device=DEV)
    prod.call("drs_conv_wgrad", xd.data_ptr(), B, S, P, cin, 0, gd.data_ptr(), P, cout, 0, k, rate, pb, cin, cin, cout, slab.data_ptr(), gw.data_ptr(),
              stream())
    torch.cuda.synchronize()
    for a, b, name in zip((out, gx, gw), res[0][:3], ("fwd", "dgrad", "wgrad")):
        np.testing.assert_array_equal(a.cpu().numpy(), b, err_msg="product library, " + name)


def test_conv1_band_padding(lib):
    """3..5 image bands ride the 32-channel K-step: padded filter rows are zero, wgrad drops them again."""
    rng = np.random.default_rng(5)
    B, S, C, cout, k = 2, 11, 5, 64, 5
    x = rng.normal(size=(B, S, S, C)).astype(np.float32)
    w = rng.normal(size=(k, k, C, cout)).astype(np.float32) * 0.1
    g = rng.normal(size=(B, S, S, cout)).astype(np.float32)
    xd = padded(x, 2, ld=32, coff=0)
    wp = torch.zeros(k * k * 32 * cout, dtype=torch.float32, device=DEV)
    wsrc = dev(w)
    lib.call("drs_filter_pad_cin", wsrc.data_ptr(), wp.data_ptr(), k, C, 32, cout, stream())
    out = torch.zeros(B * S * S * cout, dtype=torch.float32, device=DEV)
    lib.call("drs_conv_forward", xd.data_ptr(), B, S, 2, 32, 0, wp.data_ptr(), None, k, 1, 2, 32, cout, out.data_ptr(), cout, 0, 0,
             None, stream())
    gd = padded(g, 2)
    nsplit = lib.query("drs_conv_wgrad_splits", B, S, k, 32, cout)
    slab = torch.zeros(nsplit * k * k * 32 * cout, dtype=torch.float32, device=DEV)
    gw = torch.zeros(k * k * C * cout, dtype=torch.float32, device=DEV)
    lib.call("drs_conv_wgrad", xd.data_ptr(), B, S, 2, 32, 0, gd.data_ptr(), 2, cout, 0, k, 1, 2, 32, C, cout, slab.data_ptr(),
             gw.data_ptr(), stream())
    torch.cuda.synchronize()
    ref = T.conv2d_same(x.astype(np.float64), w.astype(np.float64), 1)
    _, gw_ref = T.conv2d_same_bwd(x.astype(np.float64), w.astype(np.float64), 1, g.astype(np.float64))
    assert rel_err(out.cpu().numpy().reshape(B, S, S, cout), ref) < 1e-5
    assert rel_err(gw.cpu().numpy().reshape(k, k, C, cout), gw_ref) < 1e-5


@pytest.mark.parametrize("C,cout,k,rate,B,S,CP", [(5, 64, 5, 1, 2, 11, 8), (3, 64, 5, 1, 1, 25, 8), (4, 32, 5, 1, 2, 13, 8), (5, 64, 3, 2, 2, 9, 8),
                                                     (12, 64, 5, 1, 2, 10, 16), (9, 128, 3, 3, 1, 14, 16)])
def test_conv1_packed_taps(lib, C, cout, k, rate, B, S, CP):
    """the few-band input of conv1 in an 8- (or 16-) channel slab: 32 / 8 filter taps share a K-step (k*k*8 rows, padded to a multiple of 32
    with zero filter rows) instead of one tap per 32-channel K-step; wgrad works on the same 8-channel rows."""
    rng = np.random.default_rng(C * 100 + cout + S)
    x = rng.normal(size=(B, S, S, C)).astype(np.float32)
    w = rng.normal(size=(k, k, C, cout)).astype(np.float32) * 0.1
    g = rng.normal(size=(B, S, S, cout)).astype(np.float32)
    pb, pa = onets.same_pad(k, rate)
    P = max(pb, pa)
    xd = padded(x, P, ld=CP, coff=0)
    rows = -(-k * k * CP // 32) * 32
    wp = torch.zeros(rows * cout, dtype=torch.float32, device=DEV)
    lib.call("drs_filter_pad_cin", dev(w).data_ptr(), wp.data_ptr(), k, C, CP, cout, stream())
    M = B * S * S
    out = torch.zeros(M * cout, dtype=torch.float32, device=DEV)
    mt = lib.query("drs_conv_mtile", cout)
    nrow = (M + mt - 1) // mt
    stats = torch.zeros(nrow * cout * 2, dtype=torch.float32, device=DEV)
    lib.call("drs_conv_forward", xd.data_ptr(), B, S, P, CP, 0, wp.data_ptr(), None, k, rate, pb, CP, cout, out.data_ptr(), cout, 0, 0,
             stats.data_ptr(), stream())
    gd = padded(g, P)
    nsplit = lib.query("drs_conv_wgrad_splits", B, S, k, CP, cout)
    slab = torch.zeros(nsplit * k * k * CP * cout, dtype=torch.float32, device=DEV)
    gw = torch.zeros(k * k * C * cout, dtype=torch.float32, device=DEV)
    lib.call("drs_conv_wgrad", xd.data_ptr(), B, S, P, CP, 0, gd.data_ptr(), P, cout, 0, k, rate, pb, CP, C, cout, slab.data_ptr(),
             gw.data_ptr(), stream())
    torch.cuda.synchronize()
    ref = T.conv2d_same(x.astype(np.float64), w.astype(np.float64), rate)
    _, gw_ref = T.conv2d_same_bwd(x.astype(np.float64), w.astype(np.float64), rate, g.astype(np.float64))
    assert rel_err(out.cpu().numpy().reshape(B, S, S, cout), ref) < 1e-5
    st = conv_stats_moments(lib, stats, M, mt, cout)
    assert np.abs(st[:, 0] - ref.reshape(-1, cout).sum(axis=0)).max() < 1e-5 * np.abs(ref).sum(axis=(0, 1, 2)).max()
    assert rel_err(gw.cpu().numpy().reshape(k, k, C, cout), gw_ref) < 1e-5


@pytest.mark.parametrize("C,pool,alpha,B,S,P", [(64, 1, 0.1, 2, 9, 4), (192, 1, 0.1, 1, 12, 0), (256, 0, 0.0, 2, 7, 6),
                                              (32, 0, 0.0, 3, 10, 6), (128, 1, 0.1, 2, 25, 5),
                                              # the sliding pool kernels at their edges: more columns per workgroup than the patch has
                                              # (32 channels), two columns per workgroup (448, 512 channels), ReLU's runs of exact ties at
                                              # zero, one- and two-pixel patches, a strip split (128 / 33), and the gathering form that
                                              # takes over above 512 channels
                                              (32, 1, 0.1, 2, 5, 2), (448, 1, 0.0, 1, 7, 3), (512, 1, 0.1, 1, 6, 0), (576, 1, 0.1, 1, 5, 1),
                                              (64, 1, 0.1, 3, 1, 1), (64, 1, 0.0, 2, 2, 0), (128, 1, 0.1, 12, 33, 2), (192, 1, 0.0, 3, 21, 1)])
def test_bn_act_pool_forward_backward(lib, C, pool, alpha, B, S, P):
    rng = np.random.default_rng(C + S)
    M = B * S * S
    z = (rng.normal(size=(B, S, S, C)) * 1.5 + 0.3).astype(np.float32)
    if S >= 2:
        z[0, 0, 0, :] = z[0, 0, 1, :] = 10.0   # an exact tie of two maxima inside pooling windows
    ga = rng.normal(size=(B, S, S, C)).astype(np.float32)
    kind = "relu" if alpha == 0.0 else "lrelu"
    z64 = z.astype(np.float64)
    xh, mean, var = T.batch_norm_train(z64)
    a = T.act_fwd(xh, kind)
    if pool:
        ref, idx_ref = T.max_pool_3x3(a)
    else:
        ref, idx_ref = a, None
    zd = dev(z.reshape(M, C))
    # statistics through the slab path the conv epilogue feeds
    part = np.stack([z.reshape(M, C).sum(axis=0), (z.reshape(M, C).astype(np.float64) ** 2).sum(axis=0)], axis=1).astype(np.float32)
    sums = torch.zeros(C * 2, dtype=torch.float64, device=DEV)
    partd = dev(part.reshape(1, C, 2))
    scr = torch.zeros(lib.query("drs_colsum_scratch_doubles", 2 * C), dtype=torch.float64, device=DEV)
    lib.call("drs_stats_reduce", partd.data_ptr(), 1, C, sums.data_ptr(), scr.data_ptr(), stream())
    mr = torch.zeros(C * 2, dtype=torch.float32, device=DEV)
    mm = torch.zeros(C, dtype=torch.float32, device=DEV)
    mv = torch.ones(C, dtype=torch.float32, device=DEV)
    lib.call("drs_bn_finish", sums.data_ptr(), float(M), C, mr.data_ptr(), mm.data_ptr(), mv.data_ptr(), 0.999, 1, stream())
    torch.cuda.synchronize()
    mrh = mr.cpu().numpy().reshape(C, 2)
    assert rel_err(mrh[:, 0], mean) < 1e-5 and rel_err(mrh[:, 1], 1 / np.sqrt(var + 1e-3)) < 1e-5
    assert rel_err(mm.cpu().numpy(), T.moving_update(np.zeros(C), mean)) < 1e-5
    assert rel_err(mv.cpu().numpy(), T.moving_update(np.ones(C), var * M / (M - 1.0))) < 1e-6
    ld = C + 8
    out = torch.full((B * (S + 2 * P) ** 2 * ld,), 5.0, dtype=torch.float32, device=DEV)
    idx = torch.zeros(M * C, dtype=torch.uint8, device=DEV)
    lib.call("drs_bn_act_pool_forward", zd.data_ptr(), B, S, C, mr.data_ptr(), alpha, pool, out.data_ptr(), P, ld, 4,
             idx.data_ptr() if pool else None, stream())
    torch.cuda.synchronize()
    got, full = unpad(out, B, S, P, ld, 4, C)
    assert rel_err(got, ref) < 2e-5
    if P:
        halo = full[:, :, :, 4:4 + C].copy()
        halo[:, P:P + S, P:P + S] = 0
        assert np.all(halo == 0)                           # halo of the slice zeroed
    assert np.all(full[:, :, :, :4] == 5.0) and np.all(full[:, :, :, 4 + C:] == 5.0)
    if pool:
        idx_h = idx.cpu().numpy().reshape(B, S, S, C)
        # the winner may differ from the fp64 oracle only where two fp32 candidates tie or nearly tie
        assert (idx_h != idx_ref).mean() < 1e-3
        if S >= 2:
            assert idx_h[0, 0, 0, 0] == 4 and idx_h[0, 0, 1, 0] == 3   # first maximum in scan order wins
        else:
            assert np.all(idx_h == 4)
    # backward (uses the device's own arg-max codes, checked above)
    if pool:
        idx_use = idx.cpu().numpy().reshape(B, S, S, C)
        gpool = T.max_pool_3x3_bwd(idx_use, ga.astype(np.float64))
    else:
        gpool = ga.astype(np.float64)
    gxh_ref = T.act_bwd(xh, kind, gpool)
    gz_ref = T.batch_norm_train_bwd(z64, mean, var, gxh_ref)
    gad = torch.zeros(M, C + 16, dtype=torch.float32, device=DEV)
    gad[:, 16:] = dev(ga.reshape(M, C))
    gxh = torch.zeros(M * C, dtype=torch.float32, device=DEV)
    rows = lib.query("drs_bn_backward_rows", B, S, C, pool)
    partial = torch.zeros(rows * C * 2, dtype=torch.float32, device=DEV)
    lib.call("drs_bn_backward_reduce", gad.data_ptr(), C + 16, 16, zd.data_ptr(), idx.data_ptr() if pool else None, B, S, C,
             mr.data_ptr(), alpha, pool, gxh.data_ptr(), partial.data_ptr(), stream())
    bs = torch.zeros(C * 2, dtype=torch.float64, device=DEV)
    lib.call("drs_stats_reduce", partial.data_ptr(), rows, C, bs.data_ptr(), scr.data_ptr(), stream())
    Pg = 3
    gz = torch.full((B * (S + 2 * Pg) ** 2 * C,), 9.0, dtype=torch.float32, device=DEV)
    lib.call("drs_bn_backward_apply", gxh.data_ptr(), zd.data_ptr(), B, S, C, mr.data_ptr(), bs.data_ptr(), float(M), gz.data_ptr(),
             Pg, C, 0, stream())
    torch.cuda.synchronize()
    # (the activation's derivative jumps at 0: an element whose normalised value is within fp32 rounding of zero may take the other
    #  branch on the device than in the fp64 oracle -- about one element in 10^7 -- and is left out; tests/fuzz/fuzz_pointwise.py)
    sure = np.abs(xh) > 1e-5
    assert rel_err(np.where(sure, gxh.cpu().numpy().reshape(B, S, S, C), 0.0), np.where(sure, gxh_ref, 0.0)) < 2e-5
    assert (~sure).sum() <= max(4, 1e-4 * sure.size)
    got, full = unpad(gz, B, S, Pg, C, 0, C)
    if not sure.all():          # the batch-norm backward of what the device really fed it (one flipped element moves a channel's sums by 1 / M)
        gz_ref = T.batch_norm_train_bwd(z64, mean, var, gxh.cpu().numpy().reshape(B, S, S, C).astype(np.float64))
    assert rel_err(got, gz_ref) < 5e-5
    full[:, Pg:Pg + S, Pg:Pg + S] = 0
    assert np.all(full == 0)
    # the means form (single rank: the reduction leaves the two fp32 means, the apply pass does no fp64 division): the same bits
    bs2 = torch.zeros(C * 2, dtype=torch.float64, device=DEV)
    means = torch.zeros(C * 2, dtype=torch.float32, device=DEV)
    lib.call("drs_stats_reduce_means", partial.data_ptr(), rows, C, float(M), bs2.data_ptr(), means.data_ptr(), stream())
    gz2 = torch.full((B * (S + 2 * Pg) ** 2 * C,), 9.0, dtype=torch.float32, device=DEV)
    lib.call("drs_bn_backward_apply_means", gxh.data_ptr(), zd.data_ptr(), B, S, C, mr.data_ptr(), means.data_ptr(), gz2.data_ptr(), Pg, C, 0, stream())
    torch.cuda.synchronize()
    assert torch.equal(bs2, bs) and torch.equal(gz2, gz)
    assert np.array_equal(means.cpu().numpy(), (bs.cpu().numpy() / float(M)).astype(np.float32))     # (numpy: an IEEE division, as the kernel's)


@pytest.mark.parametrize("C,alpha,B,S,P", [(64, 0.1, 2, 9, 4), (256, 0.1, 3, 25, 8), (192, 0.0, 1, 12, 0), (448, 0.0, 1, 7, 3), (32, 0.1, 2, 5, 2)])
def test_finish_folded_into_the_pooled_forward_kernel_gives_the_same_bits(lib, C, alpha, B, S, P):
    """drs_bn_finish_act_pool_forward (the multi-rank forward batch norm: the sums come back from an all-reduce and the normalising
    kernel works mean / rstd / moving averages out of them itself) against drs_bn_finish followed by drs_bn_act_pool_forward:
    output slab, positions, (mean, rstd) and both moving averages bit for bit."""
    rng = np.random.default_rng(C * 7 + S)
    M = B * S * S
    z = (rng.normal(size=(M, C)) * 2.0 - 0.4).astype(np.float32)
    zd = dev(z)
    z64 = z.astype(np.float64)
    sums = dev(np.stack([z64.sum(axis=0), (z64 ** 2).sum(axis=0)], axis=1).reshape(-1), torch.float64)
    count = float(3 * M)                                   # as if three ranks had contributed
    res = []
    for fused in (False, True):
        mr = torch.zeros(C * 2, dtype=torch.float32, device=DEV)
        mm = dev(rng.normal(size=C).astype(np.float32) * 0 + 0.25)
        mv = dev(np.full(C, 1.5, dtype=np.float32))
        out = torch.full((B * (S + 2 * P) ** 2 * C,), 7.0, dtype=torch.float32, device=DEV)
        idx = torch.zeros(M * C, dtype=torch.uint8, device=DEV)
        if fused:
            lib.call("drs_bn_finish_act_pool_forward", sums.data_ptr(), count, mr.data_ptr(), mm.data_ptr(), mv.data_ptr(), 0.999, 1, zd.data_ptr(),
                     B, S, C, alpha, 1, out.data_ptr(), P, C, 0, idx.data_ptr(), stream())
        else:
            lib.call("drs_bn_finish", sums.data_ptr(), count, C, mr.data_ptr(), mm.data_ptr(), mv.data_ptr(), 0.999, 1, stream())
            lib.call("drs_bn_act_pool_forward", zd.data_ptr(), B, S, C, mr.data_ptr(), alpha, 1, out.data_ptr(), P, C, 0, idx.data_ptr(), stream())
        torch.cuda.synchronize()
        res.append((out, idx, mr, mm, mv))
    for a, b, what in zip(res[0], res[1], ("out", "positions", "mean_rstd", "moving_mean", "moving_variance")):
        assert torch.equal(a, b), what
    assert float(res[1][3][0]) != 0.25                      # the moving averages moved
    # the folded form exists for pooled blocks only: anything else is rejected, not silently run unfused
    mr = torch.zeros(C * 2, dtype=torch.float32, device=DEV)
    out = torch.zeros(B * (S + 2 * P) ** 2 * C, dtype=torch.float32, device=DEV)
    assert lib.load().drs_bn_finish_act_pool_forward(sums.data_ptr(), count, mr.data_ptr(), None, None, 0.999, 1, zd.data_ptr(), B, S, C, alpha, 0,
                                                     out.data_ptr(), P, C, 0, None, stream()) == 1


def test_bn_eval_coeffs(lib):
    rng = np.random.default_rng(0)
    C = 192
    mm, mv = rng.normal(size=C).astype(np.float32), rng.uniform(0.5, 2, size=C).astype(np.float32)
    mr = torch.zeros(C * 2, dtype=torch.float32, device=DEV)
    mmd, mvd = dev(mm), dev(mv)
    lib.call("drs_bn_eval_coeffs", mmd.data_ptr(), mvd.data_ptr(), C, mr.data_ptr(), stream())
    h = mr.cpu().numpy().reshape(C, 2)
    np.testing.assert_array_equal(h[:, 0], mm)
    assert rel_err(h[:, 1], 1 / np.sqrt(mv.astype(np.float64) + 1e-3)) < 1e-6


@pytest.mark.parametrize("C,K,B,S,P,masked", [(256, 6, 2, 9, 0, False), (448, 2, 1, 12, 6, False), (256, 7, 2, 8, 0, True),
                                             (64, 6, 1, 31, 1, False), (448, 6, 2, 13, 6, True), (128, 4, 3, 10, 0, False), (128, 3, 1, 17, 2, False),
                                             (256, 8, 5, 21, 0, True), (192, 5, 1, 40, 1, False)])
def test_classifier_loss(lib, C, K, B, S, P, masked):
    rng = np.random.default_rng(C + K)
    M = B * S * S
    feat = rng.normal(size=(B, S, S, C)).astype(np.float32)
    w = (rng.normal(size=(C, K)) / np.sqrt(C)).astype(np.float32)
    bias = rng.normal(size=K).astype(np.float32) * 0.1
    y = rng.integers(0, K, size=(B, S, S)).astype(np.uint8)
    lm = rng.integers(0, 2, size=(B, S, S)).astype(np.uint8) if masked else None
    am = rng.integers(0, 2, size=(B, S, S)).astype(np.uint8)
    fd = padded(feat, P, fill=3.0) if P else dev(feat)
    n = float(lm.sum()) if masked else float(M)
    rows = lib.query("drs_classifier_rows", B, S)
    logits = torch.zeros(M * K, dtype=torch.float32, device=DEV)
    pred = torch.zeros(M, dtype=torch.uint8, device=DEV)
    gfeat = torch.zeros(M * C, dtype=torch.float32, device=DEV)
    dwp = torch.zeros(rows * C * K, dtype=torch.float32, device=DEV)
    dbp = torch.zeros(rows * K, dtype=torch.float32, device=DEV)
    lp = torch.zeros(rows, dtype=torch.float64, device=DEV)
    conf = torch.zeros(K * K, dtype=torch.int32, device=DEV)
    yd, amd = dev(y.reshape(-1)), dev(am.reshape(-1))
    wdev, bdev = dev(w), dev(bias)
    lmd = dev(lm.reshape(-1)) if masked else None
    lib.call("drs_classifier_loss", fd.data_ptr(), B, S, P, C, 0, C, K, wdev.data_ptr(), bdev.data_ptr(), yd.data_ptr(),
             lmd.data_ptr() if masked else None, amd.data_ptr(), 1.0 / n, logits.data_ptr(), pred.data_ptr(), gfeat.data_ptr(), C, 0,
             dwp.data_ptr(), dbp.data_ptr(), lp.data_ptr(), conf.data_ptr(), stream())
    dw = torch.zeros(C * K, dtype=torch.float32, device=DEV)
    db = torch.zeros(K, dtype=torch.float32, device=DEV)
    ls = torch.zeros(1, dtype=torch.float64, device=DEV)
    scr = torch.zeros(lib.query("drs_colsum_scratch_doubles", C * K), dtype=torch.float64, device=DEV)
    lib.call("drs_rows_reduce_f32", dwp.data_ptr(), rows, C * K, dw.data_ptr(), scr.data_ptr(), stream())
    lib.call("drs_rows_reduce_f32", dbp.data_ptr(), rows, K, db.data_ptr(), scr.data_ptr(), stream())
    lib.call("drs_sum_f64", lp.data_ptr(), rows, ls.data_ptr(), stream())
    torch.cuda.synchronize()
    f64 = feat.astype(np.float64)
    lg_ref = f64 @ w.astype(np.float64) + bias.astype(np.float64)
    ce, gl = T.softmax_ce(lg_ref, y, lm)
    lg = logits.cpu().numpy().reshape(B, S, S, K)
    assert rel_err(lg, lg_ref) < 1e-5
    ph = pred.cpu().numpy().reshape(B, S, S)
    np.testing.assert_array_equal(ph, lg.argmax(axis=3))             # first maximum of the device's own logits
    assert (ph != lg_ref.argmax(axis=3)).mean() < 1e-3
    assert abs(ls.item() / n - ce) < 1e-5 * max(1.0, abs(ce))
    assert rel_err(gfeat.cpu().numpy().reshape(B, S, S, C), gl @ w.astype(np.float64).T) < 2e-5
    assert rel_err(dw.cpu().numpy().reshape(C, K), f64.reshape(-1, C).T @ gl.reshape(-1, K)) < 2e-5
    # (the bias gradient is a sum of terms that cancel -- sum_k dL/dz_k = 0 per pixel -- so it is held to the size of what was added up)
    assert np.abs(db.cpu().numpy() - gl.reshape(-1, K).sum(axis=0)).max() < 2e-6 * np.abs(gl).reshape(-1, K).sum(axis=0).max()
    cm = np.zeros((K, K), dtype=np.int64)
    np.add.at(cm, (y[am > 0], ph[am > 0]), 1)
    np.testing.assert_array_equal(conf.cpu().numpy().reshape(K, K), cm)
    # inference form: no labels
    pred2 = torch.zeros(M, dtype=torch.uint8, device=DEV)
    lib.call("drs_classifier_loss", fd.data_ptr(), B, S, P, C, 0, C, K, wdev.data_ptr(), bdev.data_ptr(), None, None, None, 0.0,
             None, pred2.data_ptr(), None, 0, 0, None, None, None, None, stream())
    torch.cuda.synchronize()
    np.testing.assert_array_equal(pred2.cpu().numpy().reshape(B, S, S), ph)


def test_momentum_l2_confusion(lib):
    rng = np.random.default_rng(1)
    n, nd = 100003, 70001
    w = rng.normal(size=n).astype(np.float32)
    g = rng.normal(size=n).astype(np.float32)
    a = rng.normal(size=n).astype(np.float32)
    wd_, gd_, ad_ = dev(w), dev(g), dev(a)
    lib.call("drs_momentum_update", wd_.data_ptr(), gd_.data_ptr(), ad_.data_ptr(), n, nd, 0.01, 0.005, 0.9, 1.0, stream())
    sc = torch.zeros(256, dtype=torch.float64, device=DEV)
    out = torch.zeros(1, dtype=torch.float64, device=DEV)
    w2 = dev(w)
    lib.call("drs_l2_loss", w2.data_ptr(), nd, sc.data_ptr(), out.data_ptr(), stream())
    torch.cuda.synchronize()
    gg = g.astype(np.float64) + np.where(np.arange(n) < nd, 0.005 * w.astype(np.float64), 0)
    acc = 0.9 * a.astype(np.float64) + gg
    assert rel_err(ad_.cpu().numpy(), acc) < 1e-6
    assert rel_err(wd_.cpu().numpy(), w - 0.01 * acc) < 1e-6
    assert abs(out.item() - 0.5 * (w[:nd].astype(np.float64) ** 2).sum()) < 1e-9 * nd
    K = 6
    y = rng.integers(0, 7, size=50000).astype(np.uint8)      # 6 = eroded boundary, ignored (isprs:1294)
    p = rng.integers(0, K, size=50000).astype(np.uint8)
    m = rng.integers(0, 2, size=50000).astype(np.uint8)
    conf = torch.zeros(K * K, dtype=torch.int32, device=DEV)
    yd, pd_, md = dev(y), dev(p), dev(m)
    lib.call("drs_confusion", yd.data_ptr(), pd_.data_ptr(), md.data_ptr(), 50000, K, 6, conf.data_ptr(), stream())
    torch.cuda.synchronize()
    cm = np.zeros((K, K), dtype=np.int64)
    keep = (m > 0) & (y != 6)
    np.add.at(cm, (y[keep], p[keep]), 1)
    np.testing.assert_array_equal(conf.cpu().numpy().reshape(K, K), cm)


@pytest.mark.parametrize("cout,B,S,ratio", [(128, 3, 21, 500.0), (64, 2, 30, 300.0), (32, 2, 19, 1000.0), (256, 1, 40, 2000.0)])
def test_bn_statistics_survive_a_large_mean(lib, cout, B, S, ratio):
    """TensorFlow's batch_norm is two-pass (isprs:655-663): its variance does not degrade when |mean| >> std.  The conv
    epilogue's tile statistics (two-pass inside the tile, Chan combination in fp64) must not either: channels with
    |mean| / std up to `ratio`, every tile shape of the fp32 kernel, ragged last M tile, one- and two-launch forms."""
    k, rate, cin = 3, 2, 32
    rng = np.random.default_rng(int(cout + ratio))
    x = rng.normal(size=(B, S, S, cin)).astype(np.float32)
    w = (rng.normal(size=(k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)       # conv output std ~ 1
    bias = (np.linspace(-1.0, 1.0, cout) * ratio).astype(np.float32)                           # channel means up to +-ratio
    pb, pa = onets.same_pad(k, rate)
    P = max(pb, pa)
    M = B * S * S
    xd, wd, bd = padded(x, P), dev(w), dev(bias)
    z = torch.zeros(M, cout, dtype=torch.float32, device=DEV)
    mt = lib.query("drs_conv_mtile", cout)
    rows = (M + mt - 1) // mt
    assert M % mt != 0
    stats = torch.zeros(rows * cout * 2, dtype=torch.float32, device=DEV)
    lib.call("drs_conv_forward", xd.data_ptr(), B, S, P, cin, 0, wd.data_ptr(), bd.data_ptr(), k, rate, pb, cin, cout, z.data_ptr(),
             cout, 0, 0, stats.data_ptr(), stream())
    mr = torch.zeros(cout * 2, dtype=torch.float32, device=DEV)
    mm = torch.zeros(cout, dtype=torch.float32, device=DEV)
    mv = torch.ones(cout, dtype=torch.float32, device=DEV)
    lib.call("drs_conv_stats_finish", stats.data_ptr(), M, mt, cout, float(M), mr.data_ptr(), mm.data_ptr(), mv.data_ptr(), 0.999, 1, None,
             stream())
    sv = conv_stats_moments(lib, stats, M, mt, cout)
    torch.cuda.synchronize()
    z64 = z.cpu().numpy().astype(np.float64)                  # the moments of the tensor the device actually produced
    mean, var = z64.mean(axis=0), z64.var(axis=0)
    assert np.abs(mean).max() / np.sqrt(var.min()) > 0.8 * ratio
    h = mr.cpu().numpy().reshape(cout, 2).astype(np.float64)
    got_var = 1.0 / h[:, 1] ** 2 - 1e-3
    assert np.abs(h[:, 0] - mean).max() < 1e-6 * np.abs(mean).max()
    assert np.abs(got_var / var - 1.0).max() < 2e-5, np.abs(got_var / var - 1.0).max()       # the plain (sum, sum of squares) form is off by ~1e-2 here
    two_launch = sv[:, 1] / M - (sv[:, 0] / M) ** 2
    assert np.abs(two_launch / var - 1.0).max() < 2e-5
    assert rel_err(mv.cpu().numpy(), T.moving_update(np.ones(cout), var * M / (M - 1.0))) < 1e-5


@pytest.mark.parametrize("k,rate,cin,cout,B,S", [(3, 8, 64, 128, 2, 21), (4, 3, 64, 64, 3, 32), (5, 2, 32, 64, 2, 9), (3, 5, 128, 256, 1, 40),
                                                  (3, 2, 96, 128, 2, 13)])
def test_register_staged_and_lds_dma_kernel_forms_are_bitwise_equal(lib, k, rate, cin, cout, B, S):
    """The forward / input-gradient and filter-gradient tiles exist in a register-staged and an LDS-DMA form (conv_mfma.hip); the
    library picks per tile by measured speed.  K order, chunk walk and summation order are the same, so the two forms must agree
    bit for bit -- also on patch sides that are not multiples of 32 (table walk), ragged M tiles and a last chunk with pixels past
    the end (the DMA form zeroes those rows in LDS)."""
    lib = lib.dev()          # libdrs_hip_dev.so: the same sources + the A/B switches of include/drs_dev.h
    rng = np.random.default_rng(S * 13 + cout + k)
    x = rng.normal(size=(B, S, S, cin)).astype(np.float32)
    w = (rng.normal(size=(k, k, cin, cout)) * 0.05).astype(np.float32)
    g = rng.normal(size=(B, S, S, cout)).astype(np.float32)
    bias = rng.normal(size=(cout,)).astype(np.float32)
    pb, pa = onets.same_pad(k, rate)
    P = max(pb, pa)
    M = B * S * S
    xd, gd, wd, bd = padded(x, P), padded(g, P), dev(w), dev(bias)
    mt = lib.query("drs_conv_mtile", cout)
    raw = lib
    res = {}
    try:
        for v in (0, 1):
            raw.drs_debug_conv_variant(v)
            raw.drs_debug_wgrad_variant(v)
            out = torch.zeros(M * cout, dtype=torch.float32, device=DEV)
            stats = torch.zeros(((M + mt - 1) // mt) * cout * 2, dtype=torch.float32, device=DEV)
            lib.call("drs_conv_forward", xd.data_ptr(), B, S, P, cin, 0, wd.data_ptr(), bd.data_ptr(), k, rate, pb, cin, cout, out.data_ptr(), cout, 0, 0,
                     stats.data_ptr(), stream())
            nsp = lib.query("drs_conv_wgrad_splits", B, S, k, cin, cout)
            slab = torch.zeros(nsp * w.size, dtype=torch.float32, device=DEV)
            gw = torch.zeros(w.size, dtype=torch.float32, device=DEV)
            lib.call("drs_conv_wgrad", xd.data_ptr(), B, S, P, cin, 0, gd.data_ptr(), P, cout, 0, k, rate, pb, cin, cin, cout, slab.data_ptr(),
                     gw.data_ptr(), stream())
            torch.cuda.synchronize()
            res[v] = (out, stats, gw, nsp)
    finally:
        raw.drs_debug_conv_variant(-1)
        raw.drs_debug_wgrad_variant(-1)
    assert res[0][3] == res[1][3]
    for a, b in zip(res[0][:3], res[1][:3]):
        assert torch.equal(a, b)
    ref = T.conv2d_same(x.astype(np.float64), w.astype(np.float64), rate) + bias.astype(np.float64)
    assert rel_err(res[1][0].cpu().numpy().reshape(ref.shape), ref) < 1e-5
    # the product library (libdrs_hip.so) picks one of the two forms per tile: the same bits, whichever it picked
    from drs_amd import _lib as prod
    out = torch.zeros(M * cout, dtype=torch.float32, device=DEV)
    stats = torch.zeros(((M + mt - 1) // mt) * cout * 2, dtype=torch.float32, device=DEV)
    prod.call("drs_conv_forward", xd.data_ptr(), B, S, P, cin, 0, wd.data_ptr(), bd.data_ptr(), k, rate, pb, cin, cout, out.data_ptr(), cout, 0, 0,
              stats.data_ptr(), stream())
    assert prod.query("drs_conv_wgrad_splits", B, S, k, cin, cout) == res[0][3]
    slab = torch.zeros(res[0][3] * w.size, dtype=torch.float32, device=DEV)
    gw = torch.zeros(w.size, dtype=torch.float32, device=DEV)
    prod.call("drs_conv_wgrad", xd.data_ptr(), B, S, P, cin, 0, gd.data_ptr(), P, cout, 0, k, rate, pb, cin, cin, cout, slab.data_ptr(), gw.data_ptr(),
              stream())
    torch.cuda.synchronize()
    assert torch.equal(out, res[1][0]) and torch.equal(stats, res[1][1]) and torch.equal(gw, res[1][2])


@pytest.mark.parametrize("S,B", [(5, 3), (8, 5), (10, 2), (11, 3), (12, 2), (16, 3), (31, 1), (32, 2), (33, 1), (48, 1), (63, 1), (96, 1),
                                 (33, 3), (37, 3), (45, 2), (65, 2), (85, 1), (100, 1)])
def test_filter_gradient_chunk_walk_over_patch_sides(lib, S, B):
    """The filter-gradient kernels walk 32-pixel chunks with a scalar state machine (ChunkWalk) and, when the patch side is not a
    multiple of 32, per-pixel offset tables that each thread advances incrementally (below 11 pixels a side: by division).  Every
    regime -- sides below 11, not / exactly a multiple of 32, chunks spanning several rows or images, a ragged last chunk -- against
    the fp64 oracle, in both kernel forms and with the all-halo chunks skipped or not, which must also agree bit for bit."""
    lib = lib.dev()          # libdrs_hip_dev.so: the same sources + the A/B switches of include/drs_dev.h
    k, rate, cin, cout = 3, 2, 64, 128
    rng = np.random.default_rng(S * 31 + B)
    x = rng.normal(size=(B, S, S, cin)).astype(np.float32)
    g = rng.normal(size=(B, S, S, cout)).astype(np.float32)
    w = rng.normal(size=(k, k, cin, cout)).astype(np.float32)
    pb, pa = onets.same_pad(k, rate)
    P = max(pb, pa)
    xd, gd = padded(x, P), padded(g, P)
    _, gw_ref = T.conv2d_same_bwd(x.astype(np.float64), w.astype(np.float64), rate, g.astype(np.float64))
    raw = lib
    outs = []
    try:
        for variant in (0, 1):
            for skip in (0, 2):
                # (r05) the LDS-DMA form addresses sides >= 32 that are not a multiple of 32 by row SEGMENTS (scalar bases, a per-lane
                # compare only in the half chunks a row ends in; the ragged last chunk by a jump back) or, switched off, by the r04 tables
                for seg in ((0, 1) if variant == 1 else (1,)):
                    raw.drs_debug_wgrad_variant(variant)
                    raw.drs_debug_skip_taps(skip)
                    raw.drs_debug_wgrad_seg(seg)
                    nsp = lib.query("drs_conv_wgrad_splits", B, S, k, cin, cout)
                    slab = torch.zeros(nsp * w.size, dtype=torch.float32, device=DEV)
                    gw = torch.full((w.size,), 7.0, dtype=torch.float32, device=DEV)
                    lib.call("drs_conv_wgrad", xd.data_ptr(), B, S, P, cin, 0, gd.data_ptr(), P, cout, 0, k, rate, pb, cin, cin, cout, slab.data_ptr(),
                             gw.data_ptr(), stream())
                    torch.cuda.synchronize()
                    outs.append(gw)
        # the cut of the pixel dimension: by live pixels (default, above) and in equal chunk ranges -- other partial sums, same gradient
        raw.drs_debug_wgrad_variant(-1)
        raw.drs_debug_skip_taps(1)
        raw.drs_debug_wgrad_balance(0)
        nsp = lib.query("drs_conv_wgrad_splits", B, S, k, cin, cout)
        slab = torch.zeros(nsp * w.size, dtype=torch.float32, device=DEV)
        gw_eq = torch.full((w.size,), 7.0, dtype=torch.float32, device=DEV)
        lib.call("drs_conv_wgrad", xd.data_ptr(), B, S, P, cin, 0, gd.data_ptr(), P, cout, 0, k, rate, pb, cin, cin, cout, slab.data_ptr(),
                 gw_eq.data_ptr(), stream())
        torch.cuda.synchronize()
    finally:
        raw.drs_debug_wgrad_variant(-1)
        raw.drs_debug_skip_taps(1)
        raw.drs_debug_wgrad_balance(1)
        raw.drs_debug_wgrad_seg(1)
    assert rel_err(outs[0].cpu().numpy().reshape(gw_ref.shape), gw_ref) < 1e-5
    for o in outs[1:]:
        assert torch.equal(outs[0], o)
    from drs_amd import _lib as prod                  # libdrs_hip.so: its own pick of form, skip and addressing -- the same bits
    nsp = prod.query("drs_conv_wgrad_splits", B, S, k, cin, cout)
    slab = torch.zeros(nsp * w.size, dtype=torch.float32, device=DEV)
    gw = torch.full((w.size,), 7.0, dtype=torch.float32, device=DEV)
    prod.call("drs_conv_wgrad", xd.data_ptr(), B, S, P, cin, 0, gd.data_ptr(), P, cout, 0, k, rate, pb, cin, cin, cout, slab.data_ptr(), gw.data_ptr(),
              stream())
    torch.cuda.synchronize()
    assert torch.equal(gw, outs[0])
    assert rel_err(gw_eq.cpu().numpy().reshape(gw_ref.shape), gw_ref) < 1e-5
    assert float((gw_eq - outs[0]).abs().max() / outs[0].abs().max()) < 2e-6


@pytest.mark.parametrize("C,K,B,S,P", [(256, 6, 3, 21, 0), (128, 4, 2, 17, 2), (64, 7, 1, 40, 1), (192, 8, 5, 9, 0), (256, 6, 16, 25, 0)])
def test_classifier_forms_agree(lib, C, K, B, S, P):
    """The classifier block exists in a vector-ALU form (2-3 classes), a register-staged MFMA form and an LDS-DMA MFMA form (features
    brought in once, one tile ahead; up to C = 256).  The two MFMA forms run the same products in the same order: bitwise equal,
    ragged last tiles, haloed feature slabs and masks included; the vector-ALU form agrees to rounding."""
    lib = lib.dev()
    rng = np.random.default_rng(C * 3 + K + S)
    M = B * S * S
    feat = rng.normal(size=(B, S, S, C)).astype(np.float32)
    w = (rng.normal(size=(C, K)) / np.sqrt(C)).astype(np.float32)
    bias = rng.normal(size=K).astype(np.float32) * 0.1
    y = rng.integers(0, K, size=M).astype(np.uint8)
    lm = rng.integers(0, 2, size=M).astype(np.uint8)
    am = rng.integers(0, 2, size=M).astype(np.uint8)
    fd = padded(feat, P, fill=3.0) if P else dev(feat)
    rows = lib.query("drs_classifier_rows", B, S)
    wdev, bdev, yd, lmd, amd = dev(w), dev(bias), dev(y), dev(lm), dev(am)
    res = {}
    try:
        for v in (0, 2, 3):
            lib.drs_debug_cls_variant(v)
            logits = torch.zeros(M * K, dtype=torch.float32, device=DEV)
            pred = torch.zeros(M, dtype=torch.uint8, device=DEV)
            gfeat = torch.zeros(M * C, dtype=torch.float32, device=DEV)
            dwp = torch.zeros(rows * C * K, dtype=torch.float32, device=DEV)
            dbp = torch.zeros(rows * K, dtype=torch.float32, device=DEV)
            lp = torch.zeros(rows, dtype=torch.float64, device=DEV)
            conf = torch.zeros(K * K, dtype=torch.int32, device=DEV)
            lib.call("drs_classifier_loss", fd.data_ptr(), B, S, P, C, 0, C, K, wdev.data_ptr(), bdev.data_ptr(), yd.data_ptr(), lmd.data_ptr(),
                     amd.data_ptr(), 1.0 / max(1, int(lm.sum())), logits.data_ptr(), pred.data_ptr(), gfeat.data_ptr(), C, 0, dwp.data_ptr(), dbp.data_ptr(),
                     lp.data_ptr(), conf.data_ptr(), stream())
            pred2 = torch.zeros(M, dtype=torch.uint8, device=DEV)
            lib.call("drs_classifier_loss", fd.data_ptr(), B, S, P, C, 0, C, K, wdev.data_ptr(), bdev.data_ptr(), None, None, None, 0.0, None,
                     pred2.data_ptr(), None, 0, 0, None, None, None, None, stream())
            torch.cuda.synchronize()
            res[v] = (logits, pred, gfeat, dwp, dbp, lp, conf, pred2)
    finally:
        lib.drs_debug_cls_variant(1)
    for a, b in zip(res[2], res[3]):
        assert torch.equal(a, b)
    from drs_amd import _lib as prod                  # libdrs_hip.so picks the form by class count and pixel count
    logits = torch.zeros(M * K, dtype=torch.float32, device=DEV)
    pred = torch.zeros(M, dtype=torch.uint8, device=DEV)
    gfeat = torch.zeros(M * C, dtype=torch.float32, device=DEV)
    dwp = torch.zeros(rows * C * K, dtype=torch.float32, device=DEV)
    dbp = torch.zeros(rows * K, dtype=torch.float32, device=DEV)
    lp = torch.zeros(rows, dtype=torch.float64, device=DEV)
    conf = torch.zeros(K * K, dtype=torch.int32, device=DEV)
    assert prod.query("drs_classifier_rows", B, S) == rows
    prod.call("drs_classifier_loss", fd.data_ptr(), B, S, P, C, 0, C, K, wdev.data_ptr(), bdev.data_ptr(), yd.data_ptr(), lmd.data_ptr(),
              amd.data_ptr(), 1.0 / max(1, int(lm.sum())), logits.data_ptr(), pred.data_ptr(), gfeat.data_ptr(), C, 0, dwp.data_ptr(), dbp.data_ptr(),
              lp.data_ptr(), conf.data_ptr(), stream())
    torch.cuda.synchronize()
    for a, b in zip((logits, pred, gfeat, dwp, dbp, lp, conf), res[2] if K >= 4 else res[0]):
        assert torch.equal(a, b)
    assert torch.equal(res[0][1], res[2][1]) or float((res[0][1] != res[2][1]).float().mean()) < 1e-3
    assert torch.equal(res[0][6], res[2][6]) or float((res[0][6] - res[2][6]).abs().sum()) <= 2e-3 * M
    for i in (0, 2):
        assert float((res[0][i] - res[2][i]).abs().max()) <= 1e-5 * float(res[2][i].abs().max())
    assert abs(float(res[0][5].sum()) - float(res[2][5].sum())) <= 1e-6 * abs(float(res[2][5].sum()))
