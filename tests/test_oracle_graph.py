"""oracle/tf_ops.py (numpy, hand-derived backward) vs oracle/torch_ref.py (PyTorch-CPU
autograd): two independent restatements of the reference graph must agree.  CPU only."""
import numpy as np
import pytest
import torch

from oracle import tf_ops as T
from oracle.torch_ref import TorchNet
from oracle import nets

CASES = [("dilated_icpr_original", 3, 6, 2, 9), ("dilated_grsl", 5, 6, 2, 10),
         ("dilated8_grsl", 5, 6, 2, 11), ("dilated_icpr_rate6_densely", 4, 2, 3, 8),
         ("dilated_icpr_rate6_small", 3, 6, 2, 8), ("dilated_icpr_vary_rate", 3, 2, 2, 9),
         ("dilated_icpr_rate6_avgpool", 3, 6, 2, 9), ("dilated_icpr_rate6_squeeze", 3, 6, 2, 8),
         ("dilated_icpr_rate6_SE", 3, 6, 3, 8)]


def test_same_pad_table():
    assert nets.same_pad(4, 3) == (4, 5)       # asymmetric
    assert nets.same_pad(4, 4) == (6, 6)
    assert nets.same_pad(5, 2) == (4, 4)
    assert nets.same_pad(3, 8) == (8, 8)
    assert nets.same_pad(5, 1) == (2, 2)
    assert nets.same_pad(4, 2) == (3, 3)


def test_mac_per_pixel_matches_baseline_md():
    def mac(net, c, k):
        return sum(kk * kk * ci * co for (_, kk, ci, co, _) in nets.conv_specs(net, c)) + nets.NETS[nets.resolve(net)]["c_last"] * k
    assert mac("dilated_grsl_rate8", 5, 6) == 2090304
    assert mac("dilated_grsl", 5, 6) == 1389888
    assert mac("dilated_icpr_original", 3, 6) == 1386688
    assert mac("dilated_icpr_rate6_densely", 4, 2) == 816128


@pytest.mark.parametrize("net,ch,K,B,s", CASES)
def test_forward_and_grads_agree(net, ch, K, B, s):
    rng = np.random.default_rng(0)
    x = rng.normal(size=(B, s, s, ch))
    y = rng.integers(0, K, size=(B, s, s))
    o = T.OracleNet(net, ch, K, dtype=np.float64, seed=1)
    # perturb biases / moving stats so that nothing cancels by accident
    for n in o.p:
        if n.endswith("biases"):
            o.p[n] = o.p[n] + rng.normal(size=o.p[n].shape) * 0.05
        if n.endswith("moving_mean"):
            o.p[n] = rng.normal(size=o.p[n].shape) * 0.1
        if n.endswith("moving_variance"):
            o.p[n] = rng.uniform(0.5, 1.5, size=o.p[n].shape)
    t = TorchNet(net, ch, K, params=o.p, dtype=torch.float64)
    # eval-mode logits
    np.testing.assert_allclose(o.forward(x, False), t.forward(x, False).detach().numpy(), rtol=1e-9, atol=1e-10)
    # train-mode loss, logits and every gradient
    wd = 0.005
    loss_o, pred_o, g_o, logits_o = o.loss_and_grads(x, y, wd)
    loss_t, logits_t, g_t = t.grads(x, y, wd)
    assert abs(loss_o - loss_t) < 1e-10
    np.testing.assert_allclose(logits_o, logits_t, rtol=1e-8, atol=1e-9)
    for n in g_t:
        np.testing.assert_allclose(g_o[n], g_t[n], rtol=1e-6, atol=1e-9, err_msg=n)
    # moving statistics after one training forward
    tp = t.get_params()
    for n in o.p:
        if "moving" in n:
            np.testing.assert_allclose(o.p[n], tp[n], rtol=1e-9, atol=1e-12, err_msg=n)


def test_masked_loss_agrees():
    rng = np.random.default_rng(2)
    net, ch, K, B, s = "dilated_grsl", 3, 7, 2, 9
    x = rng.normal(size=(B, s, s, ch))
    y = rng.integers(0, K, size=(B, s, s))
    m = rng.integers(0, 2, size=(B, s, s)).astype(bool)
    o = T.OracleNet(net, ch, K, seed=3)
    t = TorchNet(net, ch, K, params=o.p, dtype=torch.float64)
    loss_o, _, g_o, _ = o.loss_and_grads(x, y, 0.001, mask=m)
    loss_t, _, g_t = t.grads(x, y, 0.001, mask=m)
    assert abs(loss_o - loss_t) < 1e-10
    for n in g_t:
        np.testing.assert_allclose(g_o[n], g_t[n], rtol=1e-6, atol=1e-9, err_msg=n)


def test_three_training_steps_agree():
    rng = np.random.default_rng(4)
    net, ch, K, B, s = "dilated8_grsl", 5, 6, 2, 9
    o = T.OracleNet(net, ch, K, seed=5)
    t = TorchNet(net, ch, K, params=o.p, dtype=torch.float64)
    for step in range(3):
        x = rng.normal(size=(B, s, s, ch))
        y = rng.integers(0, K, size=(B, s, s))
        lo, po = o.train_step(x, y, 0.01, 0.005)
        lt, pt = t.train_step(x, y, 0.01, 0.005)
        assert abs(lo - lt) < 1e-9
        np.testing.assert_array_equal(po, pt)
    tp = t.get_params()
    for n in o.p:
        np.testing.assert_allclose(o.p[n], tp[n], rtol=1e-7, atol=1e-10, err_msg=n)


def test_finite_difference_of_conv_weight():
    rng = np.random.default_rng(6)
    net, ch, K, B, s = "dilated_grsl", 3, 6, 1, 7
    o = T.OracleNet(net, ch, K, seed=7)
    x = rng.normal(size=(B, s, s, ch))
    y = rng.integers(0, K, size=(B, s, s))
    _, _, g, _ = o.loss_and_grads(x, y, 0.0)
    name = "conv3/weights"
    idx = (1, 2, 5, 9)
    eps = 1e-6
    base = o.p[name][idx]
    mm = {n: v.copy() for n, v in o.p.items() if "moving" in n}
    vals = []
    for d in (+eps, -eps):
        o.p[name][idx] = base + d
        l, _, _, _ = o.loss_and_grads(x, y, 0.0)
        vals.append(l)
    o.p[name][idx] = base
    fd = (vals[0] - vals[1]) / (2 * eps)
    assert abs(fd - g[name][idx]) < 1e-6 * max(1.0, abs(fd))


def test_lr_schedule_and_pool_ties():
    assert T.learning_rate(0.01, 49999, 0.5) == 0.01
    assert T.learning_rate(0.01, 50000, 0.5) == 0.005
    assert abs(T.learning_rate(0.01, 100001, 0.1) - 0.0001) < 1e-18
    x = np.zeros((1, 3, 3, 1))
    out, idx = T.max_pool_3x3(x)          # all ties: first in-bounds window element wins
    assert idx[0, 0, 0, 0] == 4 and idx[0, 1, 1, 0] == 0 and idx[0, 2, 2, 0] == 0 and idx[0, 0, 2, 0] == 3
    g = T.max_pool_3x3_bwd(idx, np.ones_like(x))
    assert g.sum() == 9 and g[0, 0, 0, 0] == 4


# --------------------------------------------------------------------------- TensorFlow's own formulation of the dilated convolution
def _space_to_batch(x, pads, r):
    """tf.space_to_batch(input, paddings, block_size): zero-pad H and W, then move the r x r phases of the padded grid into the
    batch: out[(dy * r + dx) * B + b, i, j, c] = padded[b, i * r + dy, j * r + dx, c]   (TF 1.x documentation of the op)"""
    (pt, pb), (pl, pr) = pads
    B, H, W, C = x.shape
    xp = np.zeros((B, H + pt + pb, W + pl + pr, C), dtype=x.dtype)
    xp[:, pt:pt + H, pl:pl + W] = x
    Hp, Wp = xp.shape[1:3]
    assert Hp % r == 0 and Wp % r == 0
    out = np.zeros((r * r * B, Hp // r, Wp // r, C), dtype=x.dtype)
    for dy in range(r):
        for dx in range(r):
            out[(dy * r + dx) * B:(dy * r + dx + 1) * B] = xp[:, dy::r, dx::r]
    return out


def _batch_to_space(y, crops, r, B):
    """tf.batch_to_space(input, crops, block_size): the inverse interleave, then crop"""
    (ct, cb), (cl, cr) = crops
    _, h, w, C = y.shape
    full = np.zeros((B, h * r, w * r, C), dtype=y.dtype)
    for dy in range(r):
        for dx in range(r):
            full[:, dy::r, dx::r] = y[(dy * r + dx) * B:(dy * r + dx + 1) * B]
    return full[:, ct:full.shape[1] - cb, cl:full.shape[2] - cr]


def _conv2d_valid(x, w):
    k = w.shape[0]
    B, H, W, C = x.shape
    out = np.zeros((B, H - k + 1, W - k + 1, w.shape[3]), dtype=x.dtype)
    for u in range(k):
        for v in range(k):
            out += x[:, u:u + H - k + 1, v:v + W - k + 1] @ w[u, v]
    return out


def _atrous_conv2d_same_tf(x, w, rate):
    """tf.nn.atrous_conv2d(value, filters, rate, padding='SAME') the way TensorFlow 1.x computes it (the algorithm its
    documentation and nn_ops.atrous_conv2d state): the up-sampled filter has k + (k-1)(rate-1) taps, SAME pads that extent minus
    one, the odd pixel going to the bottom / right "following the same convention as conv2d()"; the padded input is extended to a
    multiple of `rate`, space_to_batch moves the rate x rate phases into the batch, an ordinary VALID convolution runs on every
    phase, batch_to_space interleaves the phases back and crops the extension."""
    k = w.shape[0]
    H, W = x.shape[1:3]
    k_up = k + (k - 1) * (rate - 1)
    pad = k_up - 1
    pt = pl = pad // 2
    pb = pr = pad - pad // 2
    eh = (rate - (H + pt + pb) % rate) % rate
    ew = (rate - (W + pl + pr) % rate) % rate
    s2b = _space_to_batch(x, ((pt, pb + eh), (pl, pr + ew)), rate)
    y = _conv2d_valid(s2b, w)
    return _batch_to_space(y, ((0, eh), (0, ew)), rate, x.shape[0])


@pytest.mark.parametrize("k,rate,H,W", [(4, 3, 11, 13), (4, 4, 10, 9), (5, 2, 9, 12), (3, 8, 17, 20), (3, 5, 12, 7), (3, 7, 25, 25),
                                        (4, 2, 8, 8), (3, 6, 6, 5), (5, 1, 7, 9)])
def test_dilated_conv_matches_tensorflows_space_to_batch_formulation(k, rate, H, W):
    """the oracle's direct form (one shifted product per tap, SAME split computed on the dilated extent) against an independent
    restatement of the algorithm TensorFlow 1.x publishes for tf.nn.atrous_conv2d -- every (kernel, rate) pair of the three scripts'
    nets (isprs:766-777, 966-981, 1001-1020; coffee:721-841), on sides that are not multiples of the rate.  The even kernels are the
    interesting ones: 4 x 4 at rate 3 pads 4 before and 5 after."""
    rng = np.random.default_rng(k * 100 + rate)
    x = rng.normal(size=(2, H, W, 3))
    w = rng.normal(size=(k, k, 3, 4))
    got = T.conv2d_same(x, w, rate)
    ref = _atrous_conv2d_same_tf(x, w, rate)
    assert got.shape == ref.shape == (2, H, W, 4)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12)


def test_known_answers_published_with_tensorflows_api():
    """The reference holds no golden vectors for its graph half and TensorFlow cannot run here (parity of the graph half is
    UNPINNED).  What TensorFlow itself PUBLISHES with the ops the reference calls is pinned here: the worked examples and the defining
    formulas of its API documentation (restated, not copied code), as known answers for oracle/tf_ops.py.
      * tf.nn.sparse_softmax_cross_entropy_with_logits (isprs:1093), the documented example: logits [[2, -5, .5, -.1], [0, 0, 1.9, 1.4],
        [-100, 100, -100, -100]], labels [0, 3, 1] -> [0.29750752, 1.1448325, 0.]; and tf.nn.softmax_cross_entropy_with_logits' example
        logits [[4, 2, 1], [0, 5, 1]] with the one-hot row [1, 0, 0] -> 0.16984604
      * SAME padding at stride 1 (the convolution guide): pad_total = max(k_eff - 1, 0) with k_eff = k + (k - 1)(rate - 1) for
        atrous_conv2d (isprs:710), pad_before = pad_total // 2, the odd pixel after
      * tf.nn.l2_loss = sum(t ** 2) / 2 (isprs:648); tf.train.exponential_decay(staircase=True) = lr * rate ** floor(step / decay_steps)
        (isprs:1686); MomentumOptimizer: accumulation = momentum * accumulation + gradient; variable -= lr * accumulation (isprs:1687)."""
    lg = np.array([[2.0, -5.0, 0.5, -0.1], [0.0, 0.0, 1.9, 1.4], [-100.0, 100.0, -100.0, -100.0]])
    want = np.array([0.29750752, 1.1448325, 0.0])
    for i, lab in enumerate([0, 3, 1]):
        ce, g = T.softmax_ce(lg[i:i + 1].reshape(1, 1, 1, 4), np.array([[[lab]]]))
        assert abs(ce - want[i]) < 2e-7 * max(1.0, want[i]), (i, ce)
        assert abs(g.sum()) < 1e-12                                   # softmax - onehot sums to zero
    ce, _ = T.softmax_ce(lg.reshape(1, 3, 1, 4), np.array([0, 3, 1]).reshape(1, 3, 1))
    assert abs(ce - want.mean()) < 1e-7                               # isprs:1095: reduce_mean over the pixels
    ce, _ = T.softmax_ce(np.array([4.0, 2.0, 1.0]).reshape(1, 1, 1, 3), np.array([[[0]]]))
    assert abs(ce - 0.16984604) < 1e-7                                # (the documentation prints float32 results)
    for k in (1, 2, 3, 4, 5):
        for rate in range(1, 9):
            k_eff = k + (k - 1) * (rate - 1)
            pad_total = max(k_eff - 1, 0)
            assert nets.same_pad(k, rate) == (pad_total // 2, pad_total - pad_total // 2)
            x = np.zeros((1, 9, 11, 1))
            x[0, 4, 5, 0] = 1.0
            w = np.arange(1.0, k * k + 1).reshape(k, k, 1, 1)
            y = T.conv2d_same(x, w, rate)                             # an impulse: the filter, flipped, placed by the padding rule
            assert y.shape == x.shape
            pb = pad_total // 2
            for u in range(k):
                for v in range(k):
                    yy, xx = 4 - (u * rate - pb), 5 - (v * rate - pb)
                    if 0 <= yy < 9 and 0 <= xx < 11:
                        assert y[0, yy, xx, 0] == w[u, v, 0, 0]
    # tf.keras.layers.LeakyReLU's documented example at the reference's slope (isprs:620-621: tf.maximum(0.1 * x, x)): [-3, -1, 0, 2] -> [-0.3, -0.1, 0, 2]
    np.testing.assert_allclose(T.act_fwd(np.array([-3.0, -1.0, 0.0, 2.0]), "lrelu"), [-0.3, -0.1, 0.0, 2.0], rtol=0, atol=1e-15)
    np.testing.assert_array_equal(T.act_fwd(np.array([-3.0, -1.0, 0.0, 2.0]), "relu"), [0.0, 0.0, 0.0, 2.0])
    # tf.keras.layers.MaxPooling2D's documented padding='same', strides 1 example has a 2 x 2 window; for the reference's 3 x 3 window the
    # documented RULE is what is pinned: SAME pads with -inf (padding never wins), here on the same 3 x 3 input [[1..3], [4..6], [7..9]]
    y, idx = T.max_pool_3x3(np.arange(1.0, 10.0).reshape(1, 3, 3, 1) - 20.0)          # all negative: a zero pad would win everywhere
    np.testing.assert_array_equal(y[0, :, :, 0] + 20.0, [[5, 6, 6], [8, 9, 9], [8, 9, 9]])
    t = np.array([[1.0, -2.0], [3.0, 0.5]])
    o = T.OracleNet("dilated_grsl", 5, 6, seed=1)
    rng = np.random.default_rng(0)
    x = rng.normal(size=(2, 6, 6, 5))
    yl = rng.integers(0, 6, size=(2, 6, 6))
    loss, _, grads, logits = o.loss_and_grads(x, yl, 0.01)
    ce, _ = T.softmax_ce(logits, yl)
    l2 = sum(0.5 * float((o.p[n] ** 2).sum()) for n in o.p if n.endswith("/weights"))
    assert abs(loss - (ce + 0.01 * l2)) < 1e-12 and abs(0.5 * (t ** 2).sum() - 7.125) < 1e-15        # wd * l2_loss per kernel (isprs:646-651)
    assert T.learning_rate(0.01, 49999, 0.5) == 0.01 and T.learning_rate(0.01, 50000, 0.5) == 0.005 and T.learning_rate(0.01, 149999, 0.1) == 0.01 * 0.1 ** 2
    w0 = {n: o.p[n].copy() for n in grads}
    o.apply_momentum(grads, 0.02)
    o.apply_momentum(grads, 0.02)                                     # accumulation = 0.9 * g + g after the second call
    for n in grads:
        np.testing.assert_allclose(o.p[n], w0[n] - 0.02 * grads[n] - 0.02 * (0.9 * grads[n] + grads[n]), rtol=0, atol=1e-15 + 1e-13 * np.abs(w0[n]).max())
