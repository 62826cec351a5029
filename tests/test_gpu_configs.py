"""-m gpu: the BASELINE.json configurations whose sizes change the kernels' behaviour, on the HIP path against the oracle.

  * configs[2] / [3] (`uniform` over [25, 85] at batch 128, `multinomial` up to 100): patch sides above 64 are where the
    halo-tap skip (drs_common.hpp: >= 4096 workgroups / >= 2^19 pixels), the table (non-affine, S % 32 != 0) form of the
    filter-gradient kernel and ragged M tiles meet.  Convolutions at B = 128, S in {65, 77, 85} and B = 16, S = 100 are
    compared with the fp64 oracle on sampled patches (outputs are per-patch independent), the filter gradient exactly (on
    a gradient that is zero outside the sampled patches) and by adjointness (on a dense one), plus skip-on / skip-off
    bitwise equality at these sizes.
  * a `loops.train` run with distribution_type="uniform" over [25, 85] at batch 128 (isprs:1727-1737, 1757-1763).
  * configs[4]: Dilated8Pooling sliding-window inference at 64 / stride 32 against `host_ref.stitch_tile` over oracle
    forwards (interior, border and shift-back windows), and one full 6000 x 6000 property run.
"""
import random
import time

import numpy as np
import pytest
import torch

from oracle import host_ref as H
from oracle import nets as onets
from oracle import tf_ops as T

pytestmark = pytest.mark.gpu

from gpu_util import DEV, conv_stats_moments, rel_err, stream   # noqa: E402


def _padded(x, P):
    return torch.nn.functional.pad(x, (0, 0, P, P, P, P)).contiguous()


# conv8, conv3 (4x4, asymmetric pad 4/5), conv6 (Cout 192: the 128x64 tile) of Dilated8Pooling
SHAPES = [(3, 8, 256, 256), (4, 3, 64, 128), (3, 6, 192, 192)]
SIZES = [(128, 65), (128, 77), (128, 85), (16, 100)]
# DenseDilated6 (isprs:914-959): conv6 / conv5 / conv4 read channels [0, Cin) of the 448-wide concat slab, their input gradients
# ACCUMULATE into the [M][448] gradient of that slab; conv2 (Cout 32) is the 256 x 32 register-staged tile
DENSE_SHAPES = [(3, 6, 320, 128), (3, 5, 192, 128), (4, 4, 128, 64), (5, 2, 32, 32)]
DENSE_SIZES = [(128, 50), (128, 75), (16, 100)]
# Dilated6Pooling (isprs:962-993) at configs[1]'s own size
GRSL_SHAPES = [(3, 5, 128, 256), (3, 6, 256, 256)]


def _split_planes(lib, t, ns):
    p = torch.zeros(ns * t.numel(), dtype=torch.int16, device=DEV)
    lib.call("drs_split_terms", t.data_ptr(), t.numel(), ns, p.data_ptr(), stream())
    return p


def _conv_case(k, rate, cin, cout, B, S, arith="f32", ld=None):
    """forward, input gradient and filter gradient of one layer shape at (B, S) on the HIP path (either arithmetic), the input a
    channel slice [0, cin) of an `ld`-wide slab whose other channels hold junk, the input gradient ACCUMULATED into an `ld`-wide
    gradient slab when ld != cin: against the fp64 oracle on sampled patches, exactly for the filter gradient of a gradient that is
    zero outside the sampled patches, by adjointness on the dense one.

    Which binary: everything that is compared with the oracle is computed by the PRODUCT library (libdrs_hip.so, `_lib.load()`: the
    code object bench.py times) under its own launch rules.  The development library (libdrs_hip_dev.so, the same sources with
    -DDRS_DEV) is called only for the two forced forms of the halo-tap skip (every tap multiplied / the all-halo ones skipped), which
    must both give the product's bits.

    cin < 8 is conv1 (isprs:1000: 5x5, `channels` -> 64): the bands zero-padded to 8 as drs_crop_normalize writes them, the filter
    through drs_filter_pad_cin, the packed-tap kernel; no input gradient (the first layer has none)."""
    from drs_amd import _lib as prod
    devl = prod.dev()
    prod.load()
    ns = {"f32": 0, "bf16x6": 3}[arith]
    first = cin < 8
    cin_real, cin = cin, (8 if first else cin)
    ld = ld or cin
    dense = ld != cin
    M = B * S * S
    g0 = torch.Generator(device=DEV).manual_seed(1000 * k + 10 * rate + S + cin_real)
    pb, pa = onets.same_pad(k, rate)
    P = max(pb, pa)
    x = torch.randn(B, S, S, cin, device=DEV, generator=g0)
    if first:
        x[..., cin_real:] = 0
    g = torch.randn(B, S, S, cout, device=DEV, generator=g0)
    w = torch.randn(k, k, cin_real, cout, device=DEV, generator=g0) / (k * k * cin_real) ** 0.5
    bias = torch.randn(cout, device=DEV, generator=g0)
    xw = x
    if dense:
        xw = torch.full((B, S, S, ld), 3.0, device=DEV)        # junk in the channels this layer must not read
        xw[..., :cin] = x
    xp, gp = _padded(xw, P), _padded(g, P)
    st = stream()
    sample = sorted({0, B // 2, B - 1})
    gs = torch.zeros_like(g)
    gs[sample] = g[sample]
    gsp = _padded(gs, P)
    gx0 = torch.randn(M, ld, device=DEV, generator=g0) if dense else None
    wk = w                                                      # the filter as the kernels take it
    if first:
        if ns:
            pytest.skip("layer stays on the exact-fp32 kernels in every arithmetic")
        # (the packed-tap kernel walks K-steps of 32 rows: the padded filter is [round_up(k*k*8, 32)][cout], the rows past the last tap zero)
        wk = torch.zeros(-(-k * k * cin // 32) * 32 * cout, device=DEV)
        prod.call("drs_filter_pad_cin", w.data_ptr(), wk.data_ptr(), k, cin_real, cin, cout, st)
    if ns:
        if cout % 64 or cin % 32:
            pytest.skip("layer stays on the exact-fp32 kernels in every arithmetic")
        xt, gt, gst = _split_planes(prod, xp, ns), _split_planes(prod, gp, ns), _split_planes(prod, gsp, ns)
        wf = torch.zeros(ns * w.numel(), dtype=torch.int16, device=DEV)
        wd = torch.zeros(ns * w.numel(), dtype=torch.int16, device=DEV)
        split_dgrad = cin % 64 == 0
        prod.call("drs_filter_split", w.data_ptr(), k, cin, cin, cout, ns, wf.data_ptr(), wd.data_ptr() if split_dgrad else None, st)
        nsp = prod.query("drs_conv_wgrad_split_splits", B, S, k, cin, cout, P, ns)
        mt = prod.query("drs_split_conv_mtile", cout)
        assert nsp == devl.query("drs_conv_wgrad_split_splits", B, S, k, cin, cout, P, ns)
    else:
        nsp = prod.query("drs_conv_wgrad_splits", B, S, k, cin, cout)
        mt = prod.query("drs_conv_mtile", cout)
        assert nsp == devl.query("drs_conv_wgrad_splits", B, S, k, cin, cout)
    wt = torch.empty(w.numel(), device=DEV)
    if not first:
        prod.call("drs_filter_flip_transpose", w.data_ptr(), wt.data_ptr(), k, cin, cout, st)
    slab = torch.empty(nsp * wk.numel(), device=DEV)
    rows = (M + mt - 1) // mt

    def run(L):
        y = torch.empty(M, cout, device=DEV)
        stats = torch.zeros(rows * cout * 2, device=DEV)
        gx = gx0.clone() if dense else torch.zeros(M, cin, device=DEV)
        gw = torch.empty(w.numel(), device=DEV)
        gws = torch.empty(w.numel(), device=DEV)
        if ns:
            L.call("drs_conv_forward_split", xt.data_ptr(), B, S, P, ld, 0, wf.data_ptr(), bias.data_ptr(), k, rate, pb, cin, cout,
                   y.data_ptr(), cout, 0, 0, stats.data_ptr(), ns, st)
            if split_dgrad:
                L.call("drs_conv_forward_split", gt.data_ptr(), B, S, P, cout, 0, wd.data_ptr(), None, k, rate, pa, cout, cin, gx.data_ptr(),
                       ld, 0, 1 if dense else 0, None, ns, st)
            else:
                L.call("drs_conv_forward", gp.data_ptr(), B, S, P, cout, 0, wt.data_ptr(), None, k, rate, pa, cout, cin, gx.data_ptr(), ld, 0,
                       1 if dense else 0, None, st)
            L.call("drs_conv_wgrad_split", xt.data_ptr(), B, S, P, ld, 0, gt.data_ptr(), P, cout, 0, k, rate, pb, cin, cin, cout,
                   slab.data_ptr(), gw.data_ptr(), ns, st)
            L.call("drs_conv_wgrad_split", xt.data_ptr(), B, S, P, ld, 0, gst.data_ptr(), P, cout, 0, k, rate, pb, cin, cin, cout,
                   slab.data_ptr(), gws.data_ptr(), ns, st)
        else:
            L.call("drs_conv_forward", xp.data_ptr(), B, S, P, ld, 0, wk.data_ptr(), bias.data_ptr(), k, rate, pb, cin, cout, y.data_ptr(),
                   cout, 0, 0, stats.data_ptr(), st)
            if not first:
                L.call("drs_conv_forward", gp.data_ptr(), B, S, P, cout, 0, wt.data_ptr(), None, k, rate, pa, cout, cin, gx.data_ptr(), ld, 0,
                       1 if dense else 0, None, st)
            L.call("drs_conv_wgrad", xp.data_ptr(), B, S, P, ld, 0, gp.data_ptr(), P, cout, 0, k, rate, pb, cin, cin_real, cout, slab.data_ptr(),
                   gw.data_ptr(), st)
            L.call("drs_conv_wgrad", xp.data_ptr(), B, S, P, ld, 0, gsp.data_ptr(), P, cout, 0, k, rate, pb, cin, cin_real, cout, slab.data_ptr(),
                   gws.data_ptr(), st)
        torch.cuda.synchronize()
        return y, stats, gx, gw, gws

    product = run(prod)             # libdrs_hip.so under its own rules: what every oracle comparison below is made on
    forced = {}
    try:
        for mode in (0, 2):         # development library: every tap / chunk multiplied; the all-halo ones skipped
            devl.drs_debug_skip_taps(mode)
            forced[mode] = run(devl)
    finally:
        devl.drs_debug_skip_taps(1)
    for a, b, c in zip(forced[0], forced[2], product):
        assert torch.equal(a, b) and torch.equal(a, c)          # skipped products are exact zeros: bitwise neutral, in either binary
    y, stats, gx, gw, gws = product
    if dense:
        assert torch.equal(gx[:, cin:], gx0[:, cin:])           # channels beyond the slice are not touched
        gx = gx[:, :cin] - gx0[:, :cin]                         # what was accumulated (one fp32 rounding of the sum apart)
    tol = 1e-5 if not dense else 2e-5
    # (1) forward and input gradient on the sampled patches against the fp64 oracle
    w64, b64 = w.cpu().numpy().astype(np.float64), bias.cpu().numpy().astype(np.float64)
    xs, gsn = x[sample][..., :cin_real].cpu().numpy().astype(np.float64), g[sample].cpu().numpy().astype(np.float64)
    ref = T.conv2d_same(xs, w64, rate) + b64
    gx_ref, gw_ref = T.conv2d_same_bwd(xs, w64, rate, gsn)
    assert rel_err(y.view(B, S, S, cout)[sample].cpu().numpy(), ref) < 1e-5
    if not first:
        assert rel_err(gx.reshape(B, S, S, cin)[sample].cpu().numpy(), gx_ref) < tol
    # (2) the filter gradient of the sampled patches (same grid, splits and chunk walk as the dense one), exactly
    assert rel_err(gws.view(k, k, cin_real, cout).cpu().numpy(), gw_ref) < 1e-5
    # (3) dense filter gradient and dense input gradient by adjointness with the (checked) forward
    yb = y.double() - bias.double()
    a = (yb * g.reshape(M, cout).double()).sum().item()
    c = (w.reshape(-1).double() * gw.double()).sum().item()
    scale = (yb.norm() * g.double().norm()).item()
    assert abs(a - c) < 1e-6 * scale, (a, c)
    if not first:
        b = (x.reshape(M, cin).double() * gx.double()).sum().item()
        assert abs(a - b) < (1e-6 if not dense else 3e-6) * scale, (a, b)
    # (4) batch-norm statistics of the epilogue: per-tile sums reduce to the moments of y (ragged last M tile included)
    sv = conv_stats_moments(prod, stats, M, mt, cout)
    y64 = y.double()
    mean = y64.mean(0).cpu().numpy()
    var = y64.var(0, unbiased=False).cpu().numpy()
    np.testing.assert_allclose(sv[:, 0] / M, mean, rtol=0, atol=1e-6 * np.abs(mean).max() + 1e-7)
    np.testing.assert_allclose(sv[:, 1] / M - (sv[:, 0] / M) ** 2, var, rtol=2e-6)


# Dilated8Pooling, every layer (isprs:1000-1021): conv1 is the packed-tap kernel on 5 bands, conv3 the 4x4 / rate 3 asymmetric 4|5 pad
D8P_SHAPES = [(5, 1, 5, 64), (5, 2, 64, 64), (4, 3, 64, 128), (4, 4, 128, 128), (3, 5, 128, 192), (3, 6, 192, 192), (3, 7, 192, 256),
              (3, 8, 256, 256)]


@pytest.mark.parametrize("B,S", [(128, 64), (16, 64), (16, 25), (16, 50)])
@pytest.mark.parametrize("k,rate,cin,cout", D8P_SHAPES)
def test_every_dilated8_layer_at_the_headline_shape_on_the_product_library(k, rate, cin, cout, B, S):
    """VERDICT r05 item 1.  The shape bench.py's headline times -- 128 x 64 x 64: >= 4096 tiles AND S % 32 == 0, i.e. plain launches,
    full tiles first, the halo-tap skip, the wave-uniform filter-gradient form -- and one rank's share of it (16 x 64 x 64) and of
    configs[2] at a small and a middle side (16 x 25: offset tables in the filter gradient, stream-K forward; 16 x 50: row segments),
    for all eight layers of `dilated_grsl_rate8`, element-wise against oracle/tf_ops.py on sampled patches, on libdrs_hip.so."""
    _conv_case(k, rate, cin, cout, B, S, "f32")


@pytest.mark.parametrize("arith", ["f32", "bf16x6"])
@pytest.mark.parametrize("B,S", SIZES)
@pytest.mark.parametrize("k,rate,cin,cout", SHAPES)
def test_conv_above_64_matches_oracle_on_sampled_patches(k, rate, cin, cout, B, S, arith):
    _conv_case(k, rate, cin, cout, B, S, arith)


@pytest.mark.parametrize("arith", ["f32", "bf16x6"])
@pytest.mark.parametrize("B,S", DENSE_SIZES)
@pytest.mark.parametrize("k,rate,cin,cout", DENSE_SHAPES)
def test_dense_net_conv_shapes_in_the_concat_slab_match_oracle(k, rate, cin, cout, B, S, arith):
    """BASELINE configs[3] (DenseDilated6, sizes up to 100): the dense-specific paths at B = 128, S = 50 / 75 and B = 16, S = 100."""
    _conv_case(k, rate, cin, cout, B, S, arith, ld=448)


@pytest.mark.parametrize("arith", ["f32", "bf16x6"])
@pytest.mark.parametrize("k,rate,cin,cout", GRSL_SHAPES)
def test_dilated_grsl_conv5_conv6_at_config2_size(k, rate, cin, cout, arith):
    """BASELINE configs[1]: Dilated6Pooling's own layers (128 -> 256 at rate 5, 256 -> 256 at rate 6) at batch 64, 64 x 64."""
    _conv_case(k, rate, cin, cout, 64, 64, arith)


def test_dense_classifier_448_channels_two_classes_at_size():
    """the DenseDilated6 classifier (C = 448, K = 2; isprs:950-957) at B = 128, S = 50: logits / loss / gradients against fp64 on
    sampled pixels and as sums."""
    from drs_amd import _lib
    B, S, C, K = 128, 50, 448, 2
    M = B * S * S
    g0 = torch.Generator(device=DEV).manual_seed(44)
    feat = torch.randn(M, C, device=DEV, generator=g0)
    w = torch.randn(C, K, device=DEV, generator=g0) / C ** 0.5
    bias = torch.randn(K, device=DEV, generator=g0) * 0.1
    y = torch.randint(0, K, (M,), device=DEV, generator=g0, dtype=torch.int32).to(torch.uint8)
    crow = _lib.query("drs_classifier_rows", B, S)
    logits = torch.empty(M * K, device=DEV)
    pred = torch.empty(M, dtype=torch.uint8, device=DEV)
    gfeat = torch.empty(M * C, device=DEV)
    dwp, dbp = torch.empty(crow * C * K, device=DEV), torch.empty(crow * K, device=DEV)
    lp = torch.empty(crow, dtype=torch.float64, device=DEV)
    conf = torch.zeros(K * K, dtype=torch.int32, device=DEV)
    st = stream()
    _lib.call("drs_classifier_loss", feat.data_ptr(), B, S, 0, C, 0, C, K, w.data_ptr(), bias.data_ptr(), y.data_ptr(), None, None, 1.0 / M,
              logits.data_ptr(), pred.data_ptr(), gfeat.data_ptr(), C, 0, dwp.data_ptr(), dbp.data_ptr(), lp.data_ptr(), conf.data_ptr(), st)
    scr = torch.zeros(_lib.query("drs_colsum_scratch_doubles", C * K), dtype=torch.float64, device=DEV)
    dw, db = torch.empty(C * K, device=DEV), torch.empty(K, device=DEV)
    _lib.call("drs_rows_reduce_f32", dwp.data_ptr(), crow, C * K, dw.data_ptr(), scr.data_ptr(), st)
    _lib.call("drs_rows_reduce_f32", dbp.data_ptr(), crow, K, db.data_ptr(), scr.data_ptr(), st)
    torch.cuda.synchronize()
    f64, w64 = feat.double(), w.double()
    lg = f64 @ w64 + bias.double()
    assert rel_err(logits.view(M, K).cpu().numpy(), lg.cpu().numpy()) < 1e-5
    p = torch.softmax(lg, 1)
    dl = (p - torch.nn.functional.one_hot(y.long(), K)) / M
    assert rel_err(gfeat.view(M, C).cpu().numpy(), (dl @ w64.t()).cpu().numpy()) < 1e-5
    assert rel_err(dw.view(C, K).cpu().numpy(), (f64.t() @ dl).cpu().numpy()) < 1e-5
    assert rel_err(db.cpu().numpy(), dl.sum(0).cpu().numpy()) < 1e-5
    lt = torch.nn.functional.cross_entropy(lg, y.long(), reduction="sum").item()
    assert abs(lp.sum().item() - lt) < 1e-6 * lt
    margin = (lg[:, 0] - lg[:, 1]).abs() > 1e-4
    assert torch.equal(pred.long()[margin], lg.argmax(1)[margin])
    want = torch.zeros(K, K, dtype=torch.int64, device=DEV)
    want.index_put_((y.long(), pred.long()), torch.ones(M, dtype=torch.int64, device=DEV), accumulate=True)
    assert torch.equal(conf.view(K, K).long(), want)


def _hold_a_drawn_size_step_to_the_oracle(spy, net_type, channels, K, wd, batch):
    """one step of a sized loop at a DRAWN size against the CPU oracle on the very patches / variables it consumed
    (as configs 1 and 2 have for their fixed size).  Of the recorded steps the one with the fewest pixels among the sides that are
    not a multiple of 32 (the table forms of the filter gradient, stream-K forward launches) -- fp64 on the CPU, so the smallest."""
    from test_gpu_single_fixed import _torch_step_loss
    cands = [r for r in spy.steps if r["S"] % 32] or spy.steps
    rec = min(cands, key=lambda r: r["S"])
    assert rec["B"] == batch and rec["S"] >= 25
    loss_ref = _torch_step_loss(rec, net_type, channels, K, wd)
    assert abs(rec["loss"] - loss_ref) < 1e-4 * abs(loss_ref), (rec["S"], rec["loss"], loss_ref)
    assert int(rec["conf"].sum()) == int(rec["acc_mask"][:rec["B"] * rec["S"] ** 2].sum().item())
    return rec


def test_uniform_size_training_loop_config3(tmp_path, capsys, monkeypatch):
    """BASELINE configs[2]: dilated_grsl_rate8, `uniform` over [25, 85], batch 128 (isprs:1727-1737: every integer size of
    the interval can be drawn; the score arrays are indexed by size - values[0], isprs:1757-1763)."""
    from drs_amd import loops, sampling as SP
    from drs_amd.cli import init_size_scores
    from drs_amd.synthetic import make_tile
    a, b = make_tile(420, 400, 5, 6, seed=31, n_seeds=60), make_tile(300, 300, 5, 6, seed=32, n_seeds=40)
    random.seed(5)
    np.random.seed(5)
    dist = SP.create_distributions_over_classes([a[1]], 25, 10)
    tdist = SP.create_distributions_over_classes([b[1]], 25, 25)
    rot = SP.create_rotation_distribution(dist)
    values = [25, 85]
    acc, occ, chosen, probs = init_size_scores("uniform", values)
    assert len(acc) == 61 and probs is None
    steps, B = 36, 128
    out = str(tmp_path) + "/"
    from test_gpu_single_fixed import _Spy
    spy = _Spy(monkeypatch, keep=8, decisions=False)
    net = loops.train([a[0]], [a[1]], dist, rot, [b[0]], [b[1]], tdist, ["b"], 0.01, B, steps, 0.005, [0.4] * 5, [0.2] * 5, "acc",
                      "uniform", values, acc, occ, chosen, probs, 20, out, 12, "dilated_grsl_rate8", "vaihingen", "none",
                      device=DEV, val_cache_dir=str(tmp_path))
    text = capsys.readouterr().out
    assert net.global_step == steps and net.s_max == 85
    sizes = [int(t) for t in text.split("\n") if t.strip().isdigit()]
    assert len(sizes) == steps and min(sizes) >= 25 and max(sizes) <= 85
    assert max(sizes) > 64 and len(set(sizes)) > 10                     # the upper half of the interval was trained on
    want = np.bincount(np.asarray(sizes) - 25, minlength=61)
    occ_before_val = np.load(out + "patch_occur_step_%d.npy" % steps)
    np.testing.assert_array_equal(occ_before_val, want)                 # every step scored its own size, once
    score = np.load(out + "patch_acc_loss_step_%d.npy" % steps)
    assert np.all(score[want == 0] == 0) and np.all(score[want > 0] > 0) and np.all(score <= want + 1e-6)   # normalised accuracies
    assert np.all(np.isfinite(net.params.cpu().numpy()))
    losses = [float(t.split("Loss= ")[1].split()[0]) for t in text.split("\n") if "Training Minibatch" in t]
    assert len(losses) == 3 and all(np.isfinite(losses)) and losses[-1] < 1.5 * losses[0]       # sizes differ per step: no monotonic claim
    assert "Validation: Overall Accuracy=" in text
    best = int(text.split("Current patch size ")[1].split()[0])
    assert best == 25 + int(np.argmax(score / np.maximum(want, 1)))     # select_best_patch_size, isprs:549-608
    rec = _hold_a_drawn_size_step_to_the_oracle(spy, "dilated_grsl_rate8", 5, 6, 0.005, B)
    assert [r["S"] for r in spy.steps] == sizes[:8] and rec["S"] in sizes[:8]


def test_dense_multinomial_loss_training_loop_config4(tmp_path, capsys, monkeypatch):
    """BASELINE configs[3] at its own sizes: DenseDilated6 (`dilated_icpr_rate6_densely`), `multinomial` over [25, 100] with the
    listed sizes {25, 50, 75, 100} at twice the base probability (isprs:61-71), update_type=loss (score += loss * epoch / 10,
    isprs:1757-1763), 4-band tiles, 2 classes, batch 32, 24 steps: every step scores its own size once, sizes above 64 are trained
    on, the best-size rule follows the scores (`loss`: the smallest mean, unsampled sizes count as 0 and win, isprs:549-608)."""
    from drs_amd import loops, sampling as SP
    from drs_amd.cli import init_size_scores
    from drs_amd.synthetic import make_tile
    a, b = make_tile(330, 350, 4, 2, seed=41, n_seeds=50), make_tile(260, 240, 4, 2, seed=42, n_seeds=30)
    random.seed(6)
    np.random.seed(6)
    dist = SP.create_distributions_over_classes([a[1]], 25, 10, num_classes=2)
    tdist = SP.create_distributions_over_classes([b[1]], 25, 25, num_classes=2)
    rot = SP.create_rotation_distribution(dist)
    values = [25, 50, 75, 100]
    acc, occ, chosen, probs = init_size_scores("multinomial", values)
    assert len(acc) == 76 and abs(probs.sum() - 1) < 1e-12 and abs(probs[25] - 2 / 76) < 1e-12
    steps, B = 24, 32
    out = str(tmp_path) + "/"
    from test_gpu_single_fixed import _Spy
    spy = _Spy(monkeypatch, keep=6, decisions=False)
    net = loops.train([a[0]], [a[1]], dist, rot, [b[0]], [b[1]], tdist, ["b"], 0.01, B, steps, 0.001, [0.4] * 4, [0.2] * 4, "loss",
                      "multinomial", values, acc, occ, chosen, probs, 20, out, 8, "dilated_icpr_rate6_densely", "vaihingen", "none",
                      num_classes=2, device=DEV, val_cache_dir=str(tmp_path))
    text = capsys.readouterr().out
    assert net.plan.dense and net.global_step == steps and net.s_max == 100
    sizes = [int(t) for t in text.split("\n") if t.strip().isdigit()]
    assert len(sizes) == steps and min(sizes) >= 25 and max(sizes) <= 100 and max(sizes) > 64
    want = np.bincount(np.asarray(sizes) - 25, minlength=76)
    np.testing.assert_array_equal(np.load(out + "patch_occur_step_%d.npy" % steps), want)
    score = np.load(out + "patch_acc_loss_step_%d.npy" % steps)
    assert np.all(score[want == 0] == 0) and np.all(score[want > 0] > 0)          # loss * epoch_counter / 10 per drawn size
    losses = [float(t.split("Loss= ")[1].split()[0]) for t in text.split("\n") if "Training Minibatch" in t]
    assert len(losses) == 3 and all(np.isfinite(losses)) and losses[-1] < 1.5 * losses[0]
    # epoch_counter is 1 throughout (the super-batch of 3200 instances is not exhausted): score = sum of batch losses / 10
    assert 0.01 * steps < score.sum() < 1.0 * steps
    assert np.all(np.isfinite(net.params.cpu().numpy()))
    assert "Validation: Overall Accuracy=" in text
    best = int(text.split("Current patch size ")[1].split()[0])
    mean = score / np.maximum(want, 1)
    assert mean[best - 25] == mean.min() == 0.0          # the smallest mean wins, and a size never drawn has mean 0 (isprs:552)
    rec = _hold_a_drawn_size_step_to_the_oracle(spy, "dilated_icpr_rate6_densely", 4, 2, 0.001, B)
    assert [r["S"] for r in spy.steps] == sizes[:6] and rec["S"] in sizes[:6]


def test_dilated_grsl_config2_steps_are_reproducible_and_learn():
    """BASELINE configs[1]: `dilated_grsl` (Dilated6Pooling, isprs:962-993), single_fixed 64 x 64, 5 bands, batch 64, patches of a
    2048 x 2048 tile: three steps, twice -- bitwise equal (no float atomics anywhere), the loss falls, the confusion matrix counts
    every pixel."""
    from drs_amd import patches as P
    from drs_amd.net import DilatedNet
    from drs_amd.synthetic import make_tile, grid_instances
    B, S = 64, 64
    tile, lab = make_tile(2048, 2048, 5, 6, seed=1234)
    pool = P.TilePool([tile], [lab], DEV)
    inst = grid_instances(2048, 2048, S, 25, B, seed=3)
    runs = []
    for rep in range(2):
        net = DilatedNet("dilated_grsl", 5, 6, 0.005, b_max=B, s_max=S, device=DEV, seed=42)
        assert len(net.plan.layers) == 6 and net.plan.layers[5].cout == 256 and net.plan.layers[4].rate == 5
        losses = []
        for i in range(3):
            P.crop_to_net(net, pool, inst, S, [0.5] * 3, [0.1] * 3)
            out = net.train_step(B, S, 0.01)
            losses.append(net.loss_value(out["loss_parts"]))
        runs.append((losses, net.params.clone()))
        assert int(out["conf"].sum().item()) == B * S * S
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1])
    assert runs[0][0][2] < runs[0][0][0]


def _oracle_tile(o, tile, lab, S, bs, mean, std, K):
    st = H.stride_for(S)
    h, w = tile.shape[:2]
    nh, nw = H.window_counts(h, w, S, st)
    batches = []
    for i in range(-(-nh * nw // bs)):
        p, _, pos = H.create_patches_per_map(tile, lab, S, st, i, bs)
        p = p.copy()
        H.normalize_images(p, mean, std)
        batches.append((o.forward(p.astype(np.float32).astype(np.float64), False).astype(np.float32), pos))
    return H.stitch_tile(h, w, K, S, batches)


def test_config5_sliding_window_dilated8_at_64_matches_oracle():
    """BASELINE configs[4] at the real window geometry (Dilated8Pooling, 64 x 64, stride 32, isprs:1241-1284) on a mosaic
    small enough for the fp64 oracle: interior windows, windows on all four borders and the shifted-back last row / column."""
    from drs_amd import loops, patches as P
    from drs_amd.net import DilatedNet
    from drs_amd.synthetic import make_tile
    net_type, ch, K, S, bs = "dilated_grsl_rate8", 5, 6, 64, 6
    h, w = 130, 141                          # (130-64) % 32 != 0 and (141-64) % 32 != 0: both axes end on a shifted-back window
    tile, lab = make_tile(h, w, ch, K, seed=17, n_seeds=30)
    mean, std = np.array([0.5, 0.5, 0.5, 0, 0]), np.array([0.25, 0.25, 0.25, 1, 1])
    d = DilatedNet(net_type, ch, K, 0.005, b_max=bs, s_max=S, device=DEV, seed=3)
    rng = np.random.default_rng(4)
    for L in d.plan.layers:                  # non-trivial moving statistics (a trained net's eval path)
        d.set_variable(L.name + "/moving_mean", rng.normal(size=L.cout) * 0.1)
        d.set_variable(L.name + "/moving_variance", rng.uniform(0.5, 1.5, size=L.cout))
    o = T.OracleNet(net_type, ch, K, seed=3)
    for n in d.variable_names():
        o.p[n] = d.get_variable(n).astype(np.float64)
    pool = P.TilePool([tile], [lab], DEV)
    nh, nw = P.window_counts(h, w, S, 32)
    assert (nh, nw) == (4, 4)
    pos_all = P.window_positions(h, w, S, 32, 0, nh * nw)
    assert pos_all[:, 0].max() == h - S and pos_all[:, 1].max() == w - S and (h - S) % 32 and (w - S) % 32
    prob_d, occ_d, total = loops.predict_tile(d, pool, 0, S, bs, mean, std, return_sums=True)
    pred, _ = loops.predict_tile(d, pool, 0, S, bs, mean, std)
    torch.cuda.synchronize()
    assert total == 16
    prob, occur, am = _oracle_tile(o, tile, lab, S, bs, mean, std, K)
    np.testing.assert_array_equal(occ_d.view(h, w).cpu().numpy(), occur[..., 0] if occur.ndim == 3 else occur)
    got_prob = prob_d.view(h, w, K).cpu().numpy()
    assert rel_err(got_prob, prob) < 1e-3                       # north-star tolerance on summed logits
    avg = prob / occur
    srt = np.sort(avg, axis=2)
    clear = (srt[..., -1] - srt[..., -2]) > 1e-3 * np.abs(avg).max()
    assert clear.mean() > 0.95
    np.testing.assert_array_equal(pred.cpu().numpy()[clear], am[clear])


def test_config5_full_6000x6000_mosaic_properties():
    """BASELINE configs[4] at full size: 34 969 windows of 64 x 64 at stride 32 over a 6000 x 6000 x 5 mosaic.  Properties that
    do not need the oracle: the overlap counts equal the window coverage (product of the per-axis coverages, shifted-back last
    window included), every pixel is covered, the label map is the arg-max of the summed logits, a repeat is bitwise equal."""
    from drs_amd import loops, patches as P, _lib
    from drs_amd.net import DilatedNet
    n, S, Bw, K = 6000, 64, 256, 6
    g0 = torch.Generator(device=DEV).manual_seed(5)
    tile = torch.rand(n * n * 5, device=DEV, generator=g0)
    pool = P.TilePool([np.zeros((S, S, 5), dtype=np.float32)], None, DEV, dtype=np.float32)     # shell; the mosaic is made on the device
    pool.tiles, pool.labels = tile, torch.zeros(n * n, dtype=torch.uint8, device=DEV)
    pool.h, pool.w = [n], [n]
    pool.tile_h = torch.tensor([n], dtype=torch.int32, device=DEV)
    pool.tile_w = torch.tensor([n], dtype=torch.int32, device=DEV)
    net = DilatedNet("dilated_grsl_rate8", 5, K, 0.005, b_max=Bw, s_max=S, device=DEV, seed=42)
    mean, std = [0.5] * 3, [0.29] * 3
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    prob, occ, total = loops.predict_tile(net, pool, 0, S, Bw, mean, std, return_sums=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nh, nw = P.window_counts(n, n, S, 32)
    assert (nh, nw) == (187, 187) and total == 34969
    print("config 5 full size: %d windows in %.2f s = %.1f M window-pixels/s" % (total, dt, total * S * S / dt / 1e6))
    cov1 = np.zeros(n, dtype=np.int64)
    for i in range(nh):
        x0 = min(i * 32, n - S)
        cov1[x0:x0 + S] += 1
    assert cov1.min() >= 1
    c = torch.from_numpy(cov1).to(DEV)
    assert torch.equal(occ.view(n, n).long(), c[:, None] * c[None, :])
    out = torch.zeros(n * n, dtype=torch.uint8, device=DEV)
    _lib.call("drs_stitch_finalize", prob.data_ptr(), occ.data_ptr(), n, n, K, out.data_ptr(), stream())
    torch.cuda.synchronize()
    avg = prob.view(n * n, K) / occ.view(-1, 1).float()
    assert torch.equal(out.long(), avg.argmax(1))
    assert bool(torch.isfinite(prob).all())
    # the corner window is the forward pass of that window alone (a batch of one runs the stream-K cut of the convolutions, a batch
    # of 256 one workgroup per tile: the same products, summed in a different order)
    P.crop_to_net(net, pool, np.array([[0, 0, 0]]), S, mean, std)
    _, lg = net.forward(1, S)
    torch.cuda.synchronize()
    corner = prob.view(n, n, K)[:32, :32]                                   # pixels covered by the first window only
    assert float((corner - lg[0, :32, :32]).abs().max()) <= 2e-5 * float(lg.abs().max())
    P.crop_to_net(net, pool, np.concatenate([np.zeros((Bw, 1), dtype=np.int64), P.window_positions(n, n, S, 32, 0, Bw)], axis=1), S, mean, std)
    _, lgb = net.forward(Bw, S)
    torch.cuda.synchronize()
    assert torch.equal(corner, lgb[0, :32, :32])                            # ... and bit for bit the first batch's forward pass
    del avg
    prob2, occ2, _ = loops.predict_tile(net, pool, 0, S, Bw, mean, std, return_sums=True)
    torch.cuda.synchronize()
    assert torch.equal(prob, prob2) and torch.equal(occ, occ2)
