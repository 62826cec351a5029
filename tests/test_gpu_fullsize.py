"""-m gpu: BASELINE.json's full sizes (Dilated8Pooling shapes at batch 128 / 64x64x5, a 2048^2 tile), where the CPU oracle
is too slow: size-independent properties of the path instead of element-wise comparison --
adjointness <conv(x), g> = <x, dgrad(g)> = <W, wgrad(x, g)>, linearity, batch-norm moments and orthogonality of its
backward, conservation of gradient mass through the pool, per-pixel zero-sum of the softmax-CE gradient, checksum of
the confusion matrix, crop vs direct slicing, overlap-add of constant logits, bitwise reproducibility of a full step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import DEV, stream   # noqa: E402

B, S = 128, 64
M = B * S * S


def _padded(x, P):
    """device [B,S,S,C] -> haloed slab (torch ops only; this is test scaffolding, not the path under test)."""
    return torch.nn.functional.pad(x, (0, 0, P, P, P, P)).contiguous()


@pytest.mark.parametrize("k,rate,cin,cout", [(3, 8, 256, 256), (4, 3, 64, 128), (5, 2, 64, 64), (3, 5, 128, 192)])
def test_conv_adjointness_and_linearity_at_full_size(k, rate, cin, cout):
    from drs_amd import _lib
    from drs_amd.nets import same_pad
    g0 = torch.Generator(device=DEV).manual_seed(k * 100 + rate)
    pb, pa = same_pad(k, rate)
    P = max(pb, pa)
    x = torch.randn(B, S, S, cin, device=DEV, generator=g0)
    x2 = torch.randn(B, S, S, cin, device=DEV, generator=g0)
    g = torch.randn(B, S, S, cout, device=DEV, generator=g0)
    w = torch.randn(k, k, cin, cout, device=DEV, generator=g0) / (k * k * cin) ** 0.5
    xp, gp = _padded(x, P), _padded(g, P)
    st = stream()

    def conv(inp):
        out = torch.empty(M, cout, device=DEV)
        _lib.call("drs_conv_forward", inp.data_ptr(), B, S, P, cin, 0, w.data_ptr(), None, k, rate, pb, cin, cout, out.data_ptr(), cout, 0,
                  0, None, st)
        return out
    y = conv(xp)
    wt = torch.empty(k * k * cin * cout, device=DEV)
    _lib.call("drs_filter_flip_transpose", w.data_ptr(), wt.data_ptr(), k, cin, cout, st)
    gx = torch.empty(M, cin, device=DEV)
    _lib.call("drs_conv_forward", gp.data_ptr(), B, S, P, cout, 0, wt.data_ptr(), None, k, rate, pa, cout, cin, gx.data_ptr(), cin, 0, 0,
              None, st)
    ns = _lib.query("drs_conv_wgrad_splits", B, S, k, cin, cout)
    slab = torch.empty(ns * k * k * cin * cout, device=DEV)
    gw = torch.empty(k * k * cin * cout, device=DEV)
    _lib.call("drs_conv_wgrad", xp.data_ptr(), B, S, P, cin, 0, gp.data_ptr(), P, cout, 0, k, rate, pb, cin, cin, cout, slab.data_ptr(),
              gw.data_ptr(), st)
    torch.cuda.synchronize()
    a = (y.double() * g.reshape(M, cout).double()).sum().item()
    b = (x.reshape(M, cin).double() * gx.double()).sum().item()
    c = (w.reshape(-1).double() * gw.double()).sum().item()
    scale = (y.double().norm() * g.double().norm()).item()
    assert abs(a - b) < 1e-6 * scale and abs(a - c) < 1e-6 * scale, (a, b, c)
    y2 = conv(_padded(x2, P))
    y12 = conv(_padded(0.5 * x - 2.0 * x2, P))
    torch.cuda.synchronize()
    assert (y12 - (0.5 * y - 2.0 * y2)).abs().max().item() < 1e-4 * y.abs().max().item()


def test_bn_pool_and_loss_invariants_at_full_size():
    from drs_amd import _lib
    C, K = 256, 6
    g0 = torch.Generator(device=DEV).manual_seed(3)
    z = torch.randn(M, C, device=DEV, generator=g0) * 2 + 0.5
    ga = torch.randn(M, C, device=DEV, generator=g0)
    st = stream()
    mt = _lib.query("drs_conv_mtile", C)
    rows = (M + mt - 1) // mt
    part = torch.stack([z.view(rows, mt, C).sum(1), (z.view(rows, mt, C) ** 2).sum(1)], dim=2).contiguous()
    sums = torch.zeros(2 * C, dtype=torch.float64, device=DEV)
    scr = torch.zeros(_lib.query("drs_colsum_scratch_doubles", 2 * C), dtype=torch.float64, device=DEV)
    _lib.call("drs_stats_reduce", part.data_ptr(), rows, C, sums.data_ptr(), scr.data_ptr(), st)
    mr = torch.zeros(2 * C, device=DEV)
    _lib.call("drs_bn_finish", sums.data_ptr(), float(M), C, mr.data_ptr(), None, None, 0.999, 1, st)
    out = torch.empty(M * C, device=DEV)
    idx = torch.empty(M * C, dtype=torch.uint8, device=DEV)
    _lib.call("drs_bn_act_pool_forward", z.data_ptr(), B, S, C, mr.data_ptr(), 1.0, 1, out.data_ptr(), 0, C, 0, idx.data_ptr(), st)
    torch.cuda.synchronize()
    # alpha = 1 makes the activation the identity: the output is the 3x3 max of the normalised z
    xh = (z - mr.view(C, 2)[:, 0]) * mr.view(C, 2)[:, 1]
    assert abs(xh.double().mean().item()) < 1e-5 and abs(xh.double().var(unbiased=False).item() - 1.0) < 2e-3
    ref = torch.nn.functional.max_pool2d(xh.view(B, S, S, C).permute(0, 3, 1, 2), 3, 1, 1).permute(0, 2, 3, 1).reshape(M, C)
    assert torch.equal(out.view(M, C), ref)                 # same fp32 expression, same maximum: bit-exact
    assert int(idx.max().item()) <= 8
    # backward: gradient mass is conserved by the pool, and the BN backward is orthogonal to 1 and to xhat per channel
    rows_b = _lib.query("drs_bn_backward_rows", B, S, C, 1)
    gxh = torch.empty(M * C, device=DEV)
    pb = torch.empty(rows_b * C * 2, device=DEV)
    _lib.call("drs_bn_backward_reduce", ga.data_ptr(), C, 0, z.data_ptr(), idx.data_ptr(), B, S, C, mr.data_ptr(), 1.0, 1, gxh.data_ptr(),
              pb.data_ptr(), st)
    bs = torch.zeros(2 * C, dtype=torch.float64, device=DEV)
    _lib.call("drs_stats_reduce", pb.data_ptr(), rows_b, C, bs.data_ptr(), scr.data_ptr(), st)
    gz = torch.empty(M * C, device=DEV)
    _lib.call("drs_bn_backward_apply", gxh.data_ptr(), z.data_ptr(), B, S, C, mr.data_ptr(), bs.data_ptr(), float(M), gz.data_ptr(), 0, C, 0, st)
    torch.cuda.synchronize()
    assert abs(gxh.double().sum().item() - ga.double().sum().item()) < 1e-6 * ga.double().abs().sum().item()
    gzv = gz.view(M, C).double()
    assert gzv.sum(0).abs().max().item() < 1e-3 * gzv.abs().sum(0).max().item()
    assert (gzv * xh.double()).sum(0).abs().max().item() < 1e-3 * gzv.abs().sum(0).max().item()
    # classifier + loss: every pixel's logit gradient sums to zero, the confusion matrix counts every pixel once
    feat = out.view(M, C)
    w = torch.randn(C, K, device=DEV, generator=g0) / 16
    bias = torch.zeros(K, device=DEV)
    y = torch.randint(0, K, (M,), device=DEV, generator=g0, dtype=torch.int32).to(torch.uint8)
    crow = _lib.query("drs_classifier_rows", B, S)
    logits = torch.empty(M * K, device=DEV)
    pred = torch.empty(M, dtype=torch.uint8, device=DEV)
    gfeat = torch.empty(M * C, device=DEV)
    dwp, dbp = torch.empty(crow * C * K, device=DEV), torch.empty(crow * K, device=DEV)
    lp = torch.empty(crow, dtype=torch.float64, device=DEV)
    conf = torch.zeros(K * K, dtype=torch.int32, device=DEV)
    _lib.call("drs_classifier_loss", feat.data_ptr(), B, S, 0, C, 0, C, K, w.data_ptr(), bias.data_ptr(), y.data_ptr(), None, None, 1.0 / M,
              logits.data_ptr(), pred.data_ptr(), gfeat.data_ptr(), C, 0, dwp.data_ptr(), dbp.data_ptr(), lp.data_ptr(), conf.data_ptr(), st)
    db = torch.empty(K, device=DEV)
    _lib.call("drs_rows_reduce_f32", dbp.data_ptr(), crow, K, db.data_ptr(), scr.data_ptr(), st)
    torch.cuda.synchronize()
    assert int(conf.sum().item()) == M and torch.equal(conf.view(K, K).sum(1).cpu(), torch.bincount(y.long(), minlength=K).int().cpu())
    assert abs(db.double().sum().item()) < 1e-6
    assert torch.equal(pred.long(), logits.view(M, K).argmax(1))
    lt = torch.nn.functional.cross_entropy(logits.view(M, K), y.long(), reduction="sum").item()
    assert abs(lp.sum().item() - lt) < 1e-5 * lt
    # gfeat = dlogits @ w^T with rows of dlogits summing to zero  =>  gfeat @ pinv-free check through <gfeat, feat> = <dlogits, logits - bias>
    probs = torch.softmax(logits.view(M, K).double(), 1)
    dl = (probs - torch.nn.functional.one_hot(y.long(), K)) / M
    assert abs((gfeat.view(M, C).double() * feat.double()).sum().item() - (dl * logits.view(M, K).double()).sum().item()) < 1e-6


@pytest.mark.parametrize("C,P_out", [(256, 8), (64, 4)])
def test_bn_pool_backward_every_element_at_full_size(C, P_out):
    """The backward elementwise launches of a block at 128 x 64 x 64 -- pool backward + leaky-ReLU backward + the batch-norm-backward
    sums (drs_bn_backward_reduce), the means (drs_stats_reduce_means) and gz = rstd * (g - mean_g - xhat * mean_gx) written into the
    haloed slab the next convolution reads (drs_bn_backward_apply_means: the single-rank form the timed step runs; and the fp64-sums
    form behind an all-reduce) -- against the same expressions in fp64 on the device, EVERY element (isprs:655-663, 745-746 backward):
    pool winners are the device's own arg-max codes (the forward pass is checked bit-exactly above)."""
    from drs_amd import _lib
    alpha = 0.1
    g0 = torch.Generator(device=DEV).manual_seed(11 + C)
    z = torch.randn(M, C, device=DEV, generator=g0) * 1.5 + 0.3
    ga = torch.randn(M, C, device=DEV, generator=g0)
    st = stream()
    z64 = z.double()
    mu, var = z64.mean(0), z64.var(0, unbiased=False)
    mr = torch.stack([mu, (var + 1e-3).rsqrt()], dim=1).float().contiguous()            # (mean, rstd) as the forward pass leaves them
    out = torch.empty(M * C, device=DEV)
    idx = torch.empty(M * C, dtype=torch.uint8, device=DEV)
    _lib.call("drs_bn_act_pool_forward", z.data_ptr(), B, S, C, mr.data_ptr(), alpha, 1, out.data_ptr(), 0, C, 0, idx.data_ptr(), st)
    rows_b = _lib.query("drs_bn_backward_rows", B, S, C, 1)
    gxh = torch.empty(M * C, device=DEV)
    pb = torch.empty(rows_b * C * 2, device=DEV)
    _lib.call("drs_bn_backward_reduce", ga.data_ptr(), C, 0, z.data_ptr(), idx.data_ptr(), B, S, C, mr.data_ptr(), alpha, 1, gxh.data_ptr(),
              pb.data_ptr(), st)
    sums = torch.zeros(2 * C, dtype=torch.float64, device=DEV)
    means = torch.zeros(2 * C, device=DEV)
    _lib.call("drs_stats_reduce_means", pb.data_ptr(), rows_b, C, float(M), sums.data_ptr(), means.data_ptr(), st)
    Sp = S + 2 * P_out
    gz = torch.full((B, Sp, Sp, C), 7.0, device=DEV)                                   # junk in the halo: the launch must zero it
    _lib.call("drs_bn_backward_apply_means", gxh.data_ptr(), z.data_ptr(), B, S, C, mr.data_ptr(), means.data_ptr(), gz.data_ptr(), P_out, C, 0, st)
    gz2 = torch.full((B, Sp, Sp, C), 7.0, device=DEV)
    _lib.call("drs_bn_backward_apply", gxh.data_ptr(), z.data_ptr(), B, S, C, mr.data_ptr(), sums.data_ptr(), float(M), gz2.data_ptr(), P_out, C, 0, st)
    torch.cuda.synchronize()
    # fp64 on the device: route ga to the pool's winners (arg-max code = dy * 3 + dx of the 3 x 3 window), activation slope, sums, apply
    mr64 = mr.double()
    xh = (z64 - mr64[:, 0]) * mr64[:, 1]
    gp = torch.zeros(B, S + 2, S + 2, C, dtype=torch.float64, device=DEV)
    g4, i4 = ga.view(B, S, S, C).double(), idx.view(B, S, S, C)
    for code in range(9):
        dy, dx = divmod(code, 3)
        gp[:, dy:dy + S, dx:dx + S, :] += torch.where(i4 == code, g4, torch.zeros((), dtype=torch.float64, device=DEV))
    gact = gp[:, 1:-1, 1:-1, :].reshape(M, C)
    del gp
    gx_ref = gact * torch.where(xh > 0, 1.0, alpha)
    assert float((gxh.view(M, C).double() - gx_ref).abs().max()) <= 1e-6 * float(gx_ref.abs().max())
    s1, s2 = gx_ref.sum(0), (gx_ref * xh).sum(0)
    assert float((sums.view(C, 2)[:, 0] - s1).abs().max()) <= 1e-6 * float(gx_ref.abs().sum(0).max())
    assert float((sums.view(C, 2)[:, 1] - s2).abs().max()) <= 1e-6 * float((gx_ref * xh).abs().sum(0).max())
    gz_ref = mr64[:, 1] * (gx_ref - s1 / M - xh * (s2 / M))
    for got in (gz, gz2):
        inner = got[:, P_out:P_out + S, P_out:P_out + S, :].reshape(M, C).double()
        assert float((inner - gz_ref).abs().max()) <= 1e-5 * float(gz_ref.abs().max())
        assert float(got[:, :P_out].abs().max()) == 0 and float(got[:, -P_out:].abs().max()) == 0          # the halo the next convolution
        assert float(got[:, :, :P_out].abs().max()) == 0 and float(got[:, :, -P_out:].abs().max()) == 0    # reads without bounds checks
    assert torch.equal(gz, gz2)          # the two forms of the apply pass: the same bits


def test_crop_equals_direct_slicing_and_stitch_of_constant_logits():
    from drs_amd import _lib, patches as P
    from drs_amd.net import DilatedNet
    from drs_amd.synthetic import make_tile, grid_instances
    tile, lab = make_tile(2048, 2048, 5, 6, seed=1234)
    pool = P.TilePool([tile], [lab], DEV)
    net = DilatedNet("dilated_grsl_rate8", 5, 6, 0.005, b_max=B, s_max=S, device=DEV)
    inst = grid_instances(2048, 2048, S, 25, B, seed=1)
    mean, std = [0.1, 0.2, 0.3], [0.5, 0.25, 2.0]
    pos = P.crop_to_net(net, pool, inst, S, mean, std)
    torch.cuda.synchronize()
    slab, Pd, ld = net.input_slab()
    got = slab.view(B, S + 2 * Pd, S + 2 * Pd, ld)[:, Pd:Pd + S, Pd:Pd + S, :5].cpu().numpy()
    want = np.stack([tile[x:x + S, y:y + S, :] for x, y in pos]).copy()
    for c in range(3):
        want[..., c] = (want[..., c] - mean[c]) / std[c]
    np.testing.assert_array_equal(got, want.astype(np.float32))
    np.testing.assert_array_equal(net.labels[:M].cpu().numpy().reshape(B, S, S), np.stack([lab[x:x + S, y:y + S] for x, y in pos]))
    # overlap-add of constant logits over a 1024^2 region: average = the constant, counts = window coverage
    h = w = 1024
    K, st_ = 6, S // 2
    nh, nw = P.window_counts(h, w, S, st_)
    prob = torch.zeros(h * w * K, device=DEV)
    occ = torch.zeros(h * w, dtype=torch.int32, device=DEV)
    const = torch.arange(K, device=DEV, dtype=torch.float32).repeat(B * S * S)
    done = 0
    while done < nh * nw:
        n = min(B, nh * nw - done)
        _lib.call("drs_stitch_accumulate", prob.data_ptr(), occ.data_ptr(), const.data_ptr(), h, w, K, S, st_, done, n, stream())
        done += n
    out = torch.zeros(h * w, dtype=torch.uint8, device=DEV)
    _lib.call("drs_stitch_finalize", prob.data_ptr(), occ.data_ptr(), h, w, K, out.data_ptr(), stream())
    torch.cuda.synchronize()
    assert int(occ.sum().item()) == nh * nw * S * S and int(occ.min().item()) >= 1
    avg = prob.view(h * w, K) / occ.view(-1, 1)
    assert torch.allclose(avg, torch.arange(K, device=DEV, dtype=torch.float32).expand(h * w, K))
    assert bool((out == K - 1).all())


def test_full_size_training_step_is_reproducible_and_learns():
    from drs_amd import patches as P
    from drs_amd.net import DilatedNet
    from drs_amd.synthetic import make_tile, grid_instances
    tile, lab = make_tile(1024, 1024, 5, 6, seed=7)
    pool = P.TilePool([tile], [lab], DEV)
    inst = grid_instances(1024, 1024, S, 25, B, seed=2)
    runs = []
    for rep in range(2):
        net = DilatedNet("dilated_grsl_rate8", 5, 6, 0.005, b_max=B, s_max=S, device=DEV, seed=42)
        losses = []
        for i in range(3):
            P.crop_to_net(net, pool, inst, S, [0.5] * 3, [0.1] * 3)
            out = net.train_step(B, S, 0.01)
            losses.append(net.loss_value(out["loss_parts"]))
        runs.append((losses, net.params.clone()))
        assert int(out["conf"].sum().item()) == M
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1])      # bitwise, no float atomics anywhere
    assert runs[0][0][2] < runs[0][0][0]


@pytest.mark.parametrize("k,rate,cin,cout", [(3, 8, 256, 256), (4, 3, 64, 128), (5, 2, 64, 64), (3, 6, 192, 192)])
def test_full_tiles_first_launch_order_changes_no_bit(k, rate, cin, cout):
    """The plain forward / input-gradient launches of the headline size start their full tiles first and the halo-skipping tiles of
    every patch last (conv_mfma.hip lpt_tile): a launch ORDER only -- outputs and the per-tile batch-norm statistics must be the bits
    of the natural order, and both must be the bits of the launch that multiplies every tap (whose skipped products are exact zeros)."""
    from drs_amd import _lib
    from drs_amd.nets import same_pad
    d = _lib.dev()
    g0 = torch.Generator(device=DEV).manual_seed(k * 10 + rate)
    pb, pa = same_pad(k, rate)
    P = max(pb, pa)
    xp = _padded(torch.randn(B, S, S, cin, device=DEV, generator=g0), P)
    w = torch.randn(k * k * cin * cout, device=DEV, generator=g0) * 0.05
    bias = torch.randn(cout, device=DEV, generator=g0)
    mt = d.query("drs_conv_mtile", cout)
    assert d.drs_debug_conv_order(B, S, k, rate, pb, cin, cout, None, 0) > 0          # this shape does take the new order
    res = []
    for skip, lpt in ((1, 1), (1, 0), (0, 0)):
        d.drs_debug_skip_taps(2 * skip)
        d.drs_debug_conv_lpt(lpt)
        z = torch.full((M * cout,), 3.0, device=DEV)
        stats = torch.zeros((M // mt) * cout * 2, device=DEV)
        d.call("drs_conv_forward", xp.data_ptr(), B, S, P, cin, 0, w.data_ptr(), bias.data_ptr(), k, rate, pb, cin, cout, z.data_ptr(), cout, 0, 0,
               stats.data_ptr(), stream())
        torch.cuda.synchronize()
        res.append((z, stats))
    d.drs_debug_skip_taps(1)
    d.drs_debug_conv_lpt(1)
    for z, st in res[1:]:
        assert torch.equal(z, res[0][0]) and torch.equal(st, res[0][1])
    assert float(res[0][0].abs().max()) > 1.0 and torch.isfinite(res[0][0]).all()
