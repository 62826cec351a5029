"""numpy restatement of the TensorFlow-1.x ops the reference's graph is made of.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Parity status of this file: UNPINNED
(no TensorFlow here, the reference has no golden vectors for the graph half); it is
cross-checked against oracle/torch_ref.py, an independent PyTorch-CPU implementation.

Every function cites the reference call site it restates
(/root/reference/isprs_dilated_random.py unless another file is named) and states the
TF semantics it assumes.  Layout is the reference's: activations NHWC, filters HWIO.
All arithmetic runs in the dtype of the inputs (float64 for the "truth" runs).
"""
import numpy as np

from . import nets as _nets

BN_EPS = 0.001        # tf.contrib.layers.batch_norm default epsilon   (isprs:658)
BN_DECAY = 0.999      # tf.contrib.layers.batch_norm default decay     (isprs:658)
MOMENTUM = 0.9        # tf.train.MomentumOptimizer(momentum=0.9)       (isprs:1687)
LR_DECAY_STEPS = 50000  # tf.train.exponential_decay(..., 50000, f, staircase=True) (isprs:1686)


# --------------------------------------------------------------------------- conv
def conv2d_same(x, w, rate=1):
    """tf.nn.atrous_conv2d(x, w, rate, 'SAME') / tf.nn.conv2d(..., 'SAME'), stride 1.

    isprs:710-712.  Cross-correlation; zero padding split as pad_before = total//2,
    pad_after = total - pad_before with total = (k-1)*rate (extra pixel bottom/right
    for even k).
    """
    k = w.shape[0]
    pb, pa = _nets.same_pad(k, rate)
    B, H, W, C = x.shape
    xp = np.zeros((B, H + pb + pa, W + pb + pa, C), dtype=x.dtype)
    xp[:, pb:pb + H, pb:pb + W, :] = x
    out = np.zeros((B, H, W, w.shape[3]), dtype=x.dtype)
    for u in range(k):
        for v in range(k):
            out += xp[:, u * rate:u * rate + H, v * rate:v * rate + W, :] @ w[u, v]
    return out


def conv2d_same_bwd(x, w, rate, gout):
    """Gradients of conv2d_same wrt input and filter (what tf.gradients gives)."""
    k = w.shape[0]
    pb, pa = _nets.same_pad(k, rate)
    B, H, W, C = x.shape
    xp = np.zeros((B, H + pb + pa, W + pb + pa, C), dtype=x.dtype)
    xp[:, pb:pb + H, pb:pb + W, :] = x
    gxp = np.zeros_like(xp)
    gw = np.zeros_like(w)
    g2 = gout.reshape(-1, gout.shape[-1])
    for u in range(k):
        for v in range(k):
            sl = (slice(None), slice(u * rate, u * rate + H), slice(v * rate, v * rate + W), slice(None))
            gw[u, v] = xp[sl].reshape(-1, C).T @ g2
            gxp[sl] += gout @ w[u, v].T
    return gxp[:, pb:pb + H, pb:pb + W, :], gw


# --------------------------------------------------------------------------- batch norm
def batch_norm_train(z, count_scale=None):
    """tf.contrib.layers.batch_norm(is_training=True, center=False) with contrib
    defaults scale=False, epsilon=1e-3 (isprs:658-660).  Statistics over B,H,W; the
    normalisation uses the biased variance.  Returns y, mean, biased var."""
    mean = z.mean(axis=(0, 1, 2))
    var = ((z - mean) ** 2).mean(axis=(0, 1, 2))
    rstd = 1.0 / np.sqrt(var + z.dtype.type(BN_EPS))
    return (z - mean) * rstd, mean, var


def batch_norm_train_bwd(z, mean, var, gy):
    """Backward of batch_norm_train wrt z (no gamma/beta)."""
    n = z.shape[0] * z.shape[1] * z.shape[2]
    rstd = 1.0 / np.sqrt(var + z.dtype.type(BN_EPS))
    xh = (z - mean) * rstd
    s1 = gy.sum(axis=(0, 1, 2)) / n
    s2 = (gy * xh).sum(axis=(0, 1, 2)) / n
    return rstd * (gy - s1 - xh * s2)


def batch_norm_eval(z, moving_mean, moving_var):
    """is_training=False branch (isprs:661-662): moving statistics."""
    return (z - moving_mean) / np.sqrt(moving_var + z.dtype.type(BN_EPS))


def moving_update(moving, value):
    """moving_averages.assign_moving_average(zero_debias=False):
    variable -= (variable - value) * (1 - decay), updates_collections=None (isprs:659)."""
    return moving - (moving - value) * moving.dtype.type(1.0 - BN_DECAY)


# --------------------------------------------------------------------------- activation
def act_fwd(x, kind):
    """tf.nn.relu (isprs:719) or leaky_relu = tf.maximum(0.1*x, x) (isprs:620-621)."""
    if kind == "relu":
        return np.maximum(x, 0)
    return np.maximum(x.dtype.type(0.1) * x, x)


def act_bwd(x, kind, g, positive=None):
    """`positive` overrides the x > 0 decision (see OracleNet: decision-aligned comparison)."""
    slope = 0.0 if kind == "relu" else 0.1
    return np.where((x > 0) if positive is None else positive, g, g * g.dtype.type(slope))


# --------------------------------------------------------------------------- max pool
def max_pool_3x3(x, forced_idx=None):
    """tf.nn.max_pool(ksize 3x3, strides 1, SAME) (isprs:745-746, 1001).  Padding never
    wins.  Also returns the arg-max code 0..8 (window scan order, first maximum wins)
    that TF's CPU MaxPoolGrad uses to route gradients.  `forced_idx` overrides the winner
    (decision-aligned comparison, see OracleNet)."""
    B, H, W, C = x.shape
    xp = np.full((B, H + 2, W + 2, C), -np.inf, dtype=x.dtype)
    xp[:, 1:-1, 1:-1, :] = x
    stack = np.stack([xp[:, dy:dy + H, dx:dx + W, :] for dy in range(3) for dx in range(3)], axis=0)
    idx = np.argmax(stack, axis=0) if forced_idx is None else forced_idx.astype(np.int64)   # first maximum in scan order
    out = np.take_along_axis(stack, idx[None], axis=0)[0]
    return out, idx.astype(np.uint8)


def max_pool_3x3_bwd(idx, g):
    B, H, W, C = g.shape
    gp = np.zeros((B, H + 2, W + 2, C), dtype=g.dtype)
    for code in range(9):
        dy, dx = divmod(code, 3)
        gp[:, dy:dy + H, dx:dx + W, :] += np.where(idx == code, g, 0)
    return gp[:, 1:-1, 1:-1, :]


def avg_pool_same(x, k):
    """tf.nn.avg_pool(ksize k x k, strides 1, SAME) (isprs:753-758): the divisor counts only in-image pixels."""
    B, H, W, C = x.shape
    r = k // 2
    xp = np.zeros((B, H + 2 * r, W + 2 * r, C), dtype=x.dtype)
    xp[:, r:r + H, r:r + W, :] = x
    ones = np.zeros((H + 2 * r, W + 2 * r), dtype=x.dtype)
    ones[r:r + H, r:r + W] = 1
    tot = np.zeros_like(x)
    cnt = np.zeros((H, W), dtype=x.dtype)
    for dy in range(k):
        for dx in range(k):
            tot += xp[:, dy:dy + H, dx:dx + W, :]
            cnt += ones[dy:dy + H, dx:dx + W]
    return tot / cnt[None, :, :, None], cnt


def avg_pool_same_bwd(g, k, cnt):
    B, H, W, C = g.shape
    r = k // 2
    gp = np.zeros((B, H + 2 * r, W + 2 * r, C), dtype=g.dtype)
    gq = g / cnt[None, :, :, None]
    for dy in range(k):
        for dx in range(k):
            gp[:, dy:dy + H, dx:dx + W, :] += gq
    return gp[:, r:r + H, r:r + W, :]


# --------------------------------------------------------------------------- squeeze-and-excitation
def se_forward(x, w1, b1, w2, b2):
    """_squeeze_excitation_layer (isprs:682-697): reduce_mean over H,W -> _fc_layer -> relu -> _fc_layer -> sigmoid -> scale."""
    s = x.mean(axis=(1, 2))
    pre1 = s @ w1 + b1
    e1 = np.maximum(pre1, 0)
    e2 = 1.0 / (1.0 + np.exp(-(e1 @ w2 + b2)))
    return x * e2[:, None, None, :], (s, e1, e2)


def se_backward(x, state, w1, w2, gy):
    s, e1, e2 = state
    hw = x.shape[1] * x.shape[2]
    ge2 = (gy * x).sum(axis=(1, 2))
    gpre2 = ge2 * e2 * (1 - e2)
    gpre1 = (gpre2 @ w2.T) * (e1 > 0)
    gs = gpre1 @ w1.T
    gx = gy * e2[:, None, None, :] + gs[:, None, None, :] / hw
    return gx, dict(w1=s.T @ gpre1, b1=gpre1.sum(axis=0), w2=e1.T @ gpre2, b2=gpre2.sum(axis=0))


# --------------------------------------------------------------------------- loss
def softmax_ce(logits, labels, mask=None):
    """loss_def (isprs:1089-1099): mean over ALL pixels of sparse softmax-CE.
    Contest variant (contest_dilated_random.py:881-901): boolean-mask first, mean over kept.
    Returns (mean CE, dCE/dlogits)."""
    K = logits.shape[-1]
    z = logits.reshape(-1, K)
    y = labels.reshape(-1).astype(np.int64)
    zmax = z.max(axis=1, keepdims=True)
    e = np.exp(z - zmax)
    se = e.sum(axis=1, keepdims=True)
    lse = np.log(se) + zmax
    ce = lse[:, 0] - z[np.arange(z.shape[0]), y]
    p = e / se
    onehot = np.zeros_like(z)
    onehot[np.arange(z.shape[0]), y] = 1
    if mask is None:
        n = z.shape[0]
        return ce.mean(), ((p - onehot) / n).reshape(logits.shape)
    m = mask.reshape(-1).astype(bool)
    n = int(m.sum())
    g = np.where(m[:, None], (p - onehot) / n, 0)
    return ce[m].mean(), g.reshape(logits.shape)


def learning_rate(lr0, global_step, factor):
    """tf.train.exponential_decay(lr0, global_step, 50000, factor, staircase=True)
    (isprs:1686 factor 0.5; coffee:1228 / contest:1021 factor 0.1)."""
    return lr0 * factor ** (global_step // LR_DECAY_STEPS)


def xavier_uniform(rng, shape, dtype=np.float32):
    """tf.contrib.layers.xavier_initializer_conv2d (isprs:702): U(-l, l),
    l = sqrt(6 / (fan_in + fan_out)), fan = k*k*C."""
    k1, k2, ci, co = shape
    lim = np.sqrt(6.0 / (k1 * k2 * ci + k1 * k2 * co))
    return rng.uniform(-lim, lim, size=shape).astype(dtype)


# --------------------------------------------------------------------------- whole net
class OracleNet(object):
    """The reference graph for one net_type: net builders isprs:761-1033, block order
    _conv_layer isprs:700-723 (conv -> +bias -> BN -> act [-> 3x3 max-pool]), classifier
    isprs:1024-1031, loss isprs:1089-1099 (+ wd * l2_loss per kernel, isprs:646-651),
    MomentumOptimizer isprs:1685-1687.

    Decision-aligned comparison: ReLU / leaky-ReLU signs and max-pool winners are discontinuous in the
    input, so a float32 implementation and this float64 restatement can legitimately disagree on a few
    near-tie elements (|xhat| ~ 1e-7, or two window candidates equal to the last bit); each such flip moves a
    gradient entry by O(1/pixels).  ``decisions`` (one dict per conv layer with boolean ``pos`` = xhat > 0
    and uint8 ``idx`` = pool winner, both taken from the implementation under test) pins those discrete
    choices so that everything continuous can be compared tightly."""

    def __init__(self, net_type, channels, num_classes, dtype=np.float64, seed=42,
                 bessel_moving_var=True):
        self.spec = _nets.NETS[_nets.resolve(net_type)]
        self.convs = _nets.conv_specs(net_type, channels)
        self.K = num_classes
        self.dtype = np.dtype(dtype)
        self.bessel = bessel_moving_var
        rng = np.random.default_rng(seed)
        self.p = {}
        for (name, k, ci, co, r) in self.convs:
            self.p[name + "/weights"] = xavier_uniform(rng, (k, k, ci, co)).astype(dtype)
            self.p[name + "/biases"] = np.full((co,), 0.1, dtype=dtype)          # isprs:707
            self.p[name + "/moving_mean"] = np.zeros((co,), dtype=dtype)
            self.p[name + "/moving_variance"] = np.ones((co,), dtype=dtype)
        for li, scope in sorted(self.spec.get("se", {}).items()):          # _fc_layer isprs:666-679, ratio 4 isprs:1042
            C = self.convs[li][3]
            self.p[scope + "_fc1/weights"] = (rng.normal(size=(C, C // 4)) * 0.005).astype(dtype)
            self.p[scope + "_fc1/biases"] = np.full((C // 4,), 0.1, dtype=dtype)
            self.p[scope + "_fc2/weights"] = (rng.normal(size=(C // 4, C)) * 0.005).astype(dtype)
            self.p[scope + "_fc2/biases"] = np.full((C,), 0.1, dtype=dtype)
        cl = self.spec["c_last"]
        self.p["conv_classifier/weights"] = xavier_uniform(rng, (1, 1, cl, num_classes)).astype(dtype)
        self.p["conv_classifier/biases"] = np.zeros((num_classes,), dtype=dtype)  # isprs:1028
        self.mom = {k: np.zeros_like(v) for k, v in self.p.items() if self.trainable(k)}
        self.global_step = 0

    @staticmethod
    def trainable(name):
        return name.endswith("/weights") or name.endswith("/biases")

    # ---- one conv block (isprs:700-723): conv -> +bias -> BN -> activation [-> pool]
    def _block_fwd(self, li, inp, is_training, decisions):
        dt = self.dtype.type
        name, k, ci, co, r = self.convs[li]
        z = conv2d_same(inp, self.p[name + "/weights"], r) + self.p[name + "/biases"]
        if is_training:
            xh, mean, var = batch_norm_train(z)
            n = z.shape[0] * z.shape[1] * z.shape[2]
            self.p[name + "/moving_mean"] = moving_update(self.p[name + "/moving_mean"], mean)
            mv_in = var * dt(n / (n - 1.0)) if self.bessel else var
            self.p[name + "/moving_variance"] = moving_update(self.p[name + "/moving_variance"], mv_in)
        else:
            xh = batch_norm_eval(z, self.p[name + "/moving_mean"], self.p[name + "/moving_variance"])
            mean = var = None
        a = act_fwd(xh, self.spec["act"])
        dec = decisions[li] if decisions is not None else {}
        ak = self.spec.get("pools", [0] * len(self.convs))[li]
        if self.spec["pool"]:
            out, idx = max_pool_3x3(a, dec.get("idx"))
        elif ak:
            out, idx = avg_pool_same(a, ak)          # idx slot carries the divisor map
        else:
            out, idx = a, None
        self._cache[li] = (inp, z, mean, var, xh, idx, dec.get("pos"))
        return out

    def _block_bwd(self, li, gout, g):
        name, k, ci, co, r = self.convs[li]
        inp, z, mean, var, xh, idx, pos = self._cache[li]
        ak = self.spec.get("pools", [0] * len(self.convs))[li]
        ga = max_pool_3x3_bwd(idx, gout) if self.spec["pool"] else (avg_pool_same_bwd(gout, ak, idx) if ak else gout)
        gxh = act_bwd(xh, self.spec["act"], ga, pos)
        gz = batch_norm_train_bwd(z, mean, var, gxh)
        gin, gw = conv2d_same_bwd(inp, self.p[name + "/weights"], r, gz)
        g[name + "/weights"] = gw
        g[name + "/biases"] = gz.sum(axis=(0, 1, 2))
        return gin

    # ---- forward; keeps what backward needs in self._cache
    def forward(self, x, is_training, decisions=None):
        x = x.astype(self.dtype)
        self._cache = {}
        n = len(self.convs)
        if self.spec.get("squeezes"):                       # isprs:1064-1086
            cur = self._block_fwd(0, x, is_training, decisions)
            for li in range(1, n, 3):
                s1 = self._block_fwd(li, cur, is_training, decisions)
                cur = np.concatenate([self._block_fwd(li + 1, s1, is_training, decisions),
                                      self._block_fwd(li + 2, s1, is_training, decisions)], axis=3)      # isprs:737-740
        elif self.spec["dense"]:                            # isprs:921-948
            cur = self._block_fwd(0, x, is_training, decisions)
            for li in range(1, n):
                cur = np.concatenate([cur, self._block_fwd(li, cur, is_training, decisions)], axis=3)
        else:
            cur = x
            self._se_cache = {}
            for li in range(n):
                cur = self._block_fwd(li, cur, is_training, decisions)
                if li in self.spec.get("se", {}):
                    sc = self.spec["se"][li]
                    xin = cur
                    cur, state = se_forward(xin, self.p[sc + "_fc1/weights"], self.p[sc + "_fc1/biases"],
                                            self.p[sc + "_fc2/weights"], self.p[sc + "_fc2/biases"])
                    self._se_cache[li] = (xin, state)
        feat = cur
        logits = feat @ self.p["conv_classifier/weights"][0, 0] + self.p["conv_classifier/biases"]
        self.cache = (self._cache, feat)
        return logits

    def loss_and_grads(self, x, y, weight_decay, mask=None, decisions=None):
        """Returns (total loss, pred, grads dict, logits).  BN in training mode."""
        dt = self.dtype.type
        logits = self.forward(x, True, decisions)
        ce, gl = softmax_ce(logits, y, mask)
        l2 = sum(0.5 * (self.p[n] ** 2).sum() for n in self.p if n.endswith("/weights"))   # tf.nn.l2_loss
        loss = ce + weight_decay * l2
        feat = self.cache[1]
        g = {}
        C = feat.shape[-1]
        g["conv_classifier/weights"] = (feat.reshape(-1, C).T @ gl.reshape(-1, self.K)).reshape(1, 1, C, self.K)
        g["conv_classifier/biases"] = gl.reshape(-1, self.K).sum(axis=0)
        gcur = gl @ self.p["conv_classifier/weights"][0, 0].T
        n = len(self.convs)
        if self.spec.get("squeezes"):
            for li in reversed(range(1, n, 3)):
                c1 = self.convs[li + 1][3]
                gs1 = self._block_bwd(li + 2, gcur[..., c1:], g) + self._block_bwd(li + 1, gcur[..., :c1], g)
                gcur = self._block_bwd(li, gs1, g)
            self._block_bwd(0, gcur, g)
        elif self.spec["dense"]:
            for li in reversed(range(1, n)):
                co = self.convs[li][3]
                gcur = gcur[..., :gcur.shape[-1] - co] + self._block_bwd(li, gcur[..., gcur.shape[-1] - co:], g)
            self._block_bwd(0, gcur, g)
        else:
            for li in reversed(range(n)):
                if li in self.spec.get("se", {}):
                    sc = self.spec["se"][li]
                    xin, state = self._se_cache[li]
                    gcur, gp = se_backward(xin, state, self.p[sc + "_fc1/weights"], self.p[sc + "_fc2/weights"], gcur)
                    g[sc + "_fc1/weights"], g[sc + "_fc1/biases"] = gp["w1"], gp["b1"]
                    g[sc + "_fc2/weights"], g[sc + "_fc2/biases"] = gp["w2"], gp["b2"]
                gcur = self._block_bwd(li, gcur, g)
        for nm in g:
            if nm.endswith("/weights"):
                g[nm] = g[nm] + dt(weight_decay) * self.p[nm]
        pred = logits.argmax(axis=3)
        return loss, pred, g, logits

    def apply_momentum(self, grads, lr):
        """ApplyMomentum, use_nesterov=False: accum = accum*momentum + grad; var -= lr*accum."""
        dt = self.dtype.type
        for n, gr in grads.items():
            self.mom[n] = self.mom[n] * dt(MOMENTUM) + gr
            self.p[n] = self.p[n] - dt(lr) * self.mom[n]
        self.global_step += 1

    def train_step(self, x, y, lr0, weight_decay, lr_factor=0.5, mask=None, decisions=None):
        """sess.run([optimizer, loss, pred_up]) with is_training=True (isprs:1750-1752)."""
        lr = learning_rate(lr0, self.global_step, lr_factor)
        loss, pred, g, logits = self.loss_and_grads(x, y, weight_decay, mask, decisions)
        self.apply_momentum(g, lr)
        return loss, pred
