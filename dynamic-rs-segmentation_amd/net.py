"""Device-side net: what every form of it shares -- parameters under TensorFlow's scope names, initialisation, feeding, checkpoints.

Host mirror of what one `sess.run` evaluates in the reference
(/root/reference/isprs_dilated_random.py): the `_conv_layer` block :700-723 per layer, the classifier
:1024-1031, `loss_def` :1089-1099, `MomentumOptimizer(...).minimize` :1685-1687 and `tf.argmax` :1690.
Three call shapes exist in the reference and are kept:

    train  sess.run([optimizer, loss, pred_up], is_training=True)    isprs:1750-1752  -> DilatedNet.train_step
    infer  sess.run([pred_up, logits],         is_training=False)   isprs:1274-1275  -> DilatedNet.forward
    val    sess.run(pred_up,                   is_training=False)   isprs:1588       -> DilatedNet.forward

There is no autograd graph: the net is a fixed chain, so the backward pass is an explicit reverse loop.
PyTorch supplies device memory, the stream and (through `comm`) the collectives; all arithmetic is in
libdrs_hip.so (include/drs.h).

`DilatedNet(...)` returns one of two subclasses: `engine.EngineNet` (the product path: one library call per sess.run, the launch
sequence lives in csrc/engine.hip) or `oplevel.OpLevelNet` (the same sequence spelled out op by op in Python: the mirror the tests
hold the engine bitwise equal to, and the host of the opt-in split-bf16 arithmetics).  This class holds neither sequence.
"""
import math
import os

import numpy as np
import torch

from . import _lib
from .nets import Plan

# arithmetic of the convolution kernels: exact fp32 MFMA (default), or fp32 operands carried as 2 / 3 bf16 terms with
# 3 / 6 partial products on the bf16 MFMA pipe (csrc/conv_split.hip); value = number of terms
ARITH_TERMS = {"f32": 0, "bf16x3": 2, "bf16x6": 3}
BN_DECAY = 0.999        # tf.contrib.layers.batch_norm default (isprs:658)
MOMENTUM = 0.9          # isprs:1687
LR_DECAY_STEPS = 50000  # isprs:1686


def _ptr(t):
    return None if t is None else t.data_ptr()


class KernelTimer(object):
    """HIP-event timing of individual launches on the stream the kernels are enqueued on (torch's current
    stream).  Used by bench.py for the per-kernel roofline figures; off in normal runs."""

    def __init__(self):
        self.records = []        # (kind, algorithmic work, start event, end event)

    def add(self, kind, work, e0, e1):
        self.records.append((kind, work, e0, e1))

    def summary(self):
        """kind -> dict(launches, ms (sum), work (sum)); synchronises."""
        torch.cuda.synchronize()
        out = {}
        for kind, work, e0, e1 in self.records:
            d = out.setdefault(kind, dict(launches=0, ms=0.0, work=0.0))
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["work"] += work
        return out


class NoComm(object):
    """Single-process stand-in for the collective interface (see dist.py)."""
    world = 1
    rank = 0
    collective = False

    def all_reduce_sum(self, t):
        return t

    def all_reduce_sum_async(self, t):
        return None

    @staticmethod
    def wait(handles):
        pass

    sync_rng = False

    @staticmethod
    def broadcast_object(obj, src=0):
        return obj

    @staticmethod
    def agree(values, what="value"):
        pass

    @staticmethod
    def all_true(flag):
        return bool(flag)

    @staticmethod
    def barrier():
        pass


def initial_params(plan, seed):
    """the flat fp32 parameter buffer at step 0 (host code): Xavier-uniform kernels (isprs:702), _fc_layer kernels truncated normal
    stddev 0.005 (isprs:669), biases 0.1 (isprs:707), classifier bias 0 (isprs:1028)"""
    rng = np.random.default_rng(seed)
    host = np.zeros(plan.n_params, dtype=np.float32)
    for name, (off, shape) in plan.offsets.items():
        n = int(np.prod(shape))
        if name.endswith("/weights") and len(shape) == 2:
            v = rng.normal(0.0, 0.005, size=n)
            while np.any(np.abs(v) > 0.01):
                bad = np.abs(v) > 0.01
                v[bad] = rng.normal(0.0, 0.005, size=int(bad.sum()))
            host[off:off + n] = v.astype(np.float32)
        elif name.endswith("/weights"):
            k1, k2, ci, co = shape
            lim = math.sqrt(6.0 / (k1 * k2 * ci + k1 * k2 * co))
            host[off:off + n] = rng.uniform(-lim, lim, size=n).astype(np.float32)
        elif name != "conv_classifier/biases":
            host[off:off + n] = 0.1
    return host


class DilatedNet(object):
    def __new__(cls, *args, **kw):
        """`DilatedNet(...)` is the step-level net (engine.EngineNet: one library call per sess.run, csrc/engine.hip) for the
        exact-fp32 arithmetic, and the op-level mirror (oplevel.OpLevelNet: the same launch sequence spelled out in Python) with
        engine=False, with DRS_OP_LEVEL=1 in the environment, or for the split-bf16 arithmetics."""
        if cls is DilatedNet:
            arith = kw.get("arith", args[11] if len(args) > 11 else "f32")
            eng = kw.get("engine")
            if eng is None:
                eng = arith == "f32" and os.environ.get("DRS_OP_LEVEL") != "1"
            if eng:
                from .engine import EngineNet
                return object.__new__(EngineNet)
            from .oplevel import OpLevelNet
            return object.__new__(OpLevelNet)
        return object.__new__(cls)

    def __init__(self, net_type, channels, num_classes, weight_decay, b_max, s_max, device="cuda:0", seed=42,
                 comm=None, bessel_moving_var=True, lr_decay_factor=0.5, arith="f32", engine=None):
        _lib.load()                       # fail loudly here if the HIP library is absent
        if arith not in ARITH_TERMS:
            raise ValueError("arith must be one of %s" % sorted(ARITH_TERMS))
        self.arith, self.ns = arith, ARITH_TERMS[arith]
        # (the few-band first block stays on the exact-fp32 kernels in every arithmetic: its packed K-steps multiply 8 padded bands where the
        # split kernels would multiply 32 -- 0.13 against 0.34 ms forward, 0.19 against 0.50 ms filter gradient at B = 128)
        self.dev = torch.device(device)
        if self.dev.type == "cuda" and self.dev.index is None and torch.cuda.is_available():
            self.dev = torch.device("cuda", torch.cuda.current_device())       # 'cuda' = the process's current device
        self.wd = float(weight_decay)
        self.b_max, self.s_max = int(b_max), int(s_max)
        self.comm = comm if comm is not None else NoComm()
        self.bessel = 1 if bessel_moving_var else 0
        self.lr_decay_factor = lr_decay_factor      # 0.5 isprs:1686; 0.1 coffee:1228, contest:1021
        self.global_step = 0
        self.debug = None
        self.timer = None
        if self.b_max * self.s_max * self.s_max >= (1 << 24):
            raise ValueError("B*S*S must stay below 2^24")
        # One process drives ONE GPU: the library sizes its stream-K workspace and cuts its launches by the CURRENT device's CU
        # count (conv_mfma.hip cu_count) and every launch of a step goes to a stream of this net's device, so that device is made -- and
        # stays -- the process's current one.  (An index-less 'cuda' was resolved above: set_device refuses it)
        if self.dev.type == "cuda" and torch.cuda.is_available():
            torch.cuda.set_device(self.dev.index)
        self.plan = Plan(net_type, channels, num_classes, first_cin_pad=8)
        self._alloc_params()
        self._init_params(seed)
        self._alloc()

    # ------------------------------------------------------------------ parameters
    def _init_params(self, seed):
        """Xavier-uniform kernels (isprs:702), biases 0.1 (isprs:707), classifier bias 0 (isprs:1028),
        moving mean 0 / variance 1 (contrib batch_norm initialisers)."""
        self.params.copy_(torch.from_numpy(initial_params(self.plan, seed)))
        bn = np.zeros(self.plan.n_bn, dtype=np.float32)
        for L in self.plan.layers:
            o = self.plan.bn_offsets[L.name]
            bn[o + L.cout:o + 2 * L.cout] = 1.0
        self.bn.copy_(torch.from_numpy(bn))

    def variable_names(self):
        names = list(self.plan.offsets)
        for L in self.plan.layers:
            names += [L.name + "/moving_mean", L.name + "/moving_variance"]
        return names

    def _slice(self, flat, name):
        if name.endswith("/moving_mean") or name.endswith("/moving_variance"):
            lname = name.rsplit("/", 1)[0]
            L = [l for l in self.plan.layers if l.name == lname][0]
            o = self.plan.bn_offsets[lname] + (L.cout if name.endswith("variance") else 0)
            return self.bn[o:o + L.cout], (L.cout,)
        off, shape = self.plan.offsets[name]
        return flat[off:off + int(np.prod(shape))], shape

    def get_variable(self, name, slot=None):
        """TF-scope-named access (`conv1/weights`, `conv1/moving_mean`, ...); slot='Momentum' reads the
        optimizer accumulator (what tf.train.Saver would store, isprs:1693-1695)."""
        flat = self.mom if slot == "Momentum" else self.params
        t, shape = self._slice(flat, name)
        return t.detach().cpu().numpy().reshape(shape).copy()

    def set_variable(self, name, value, slot=None):
        flat = self.mom if slot == "Momentum" else self.params
        t, shape = self._slice(flat, name)
        v = np.ascontiguousarray(np.asarray(value, dtype=np.float32).reshape(shape))
        t.copy_(torch.from_numpy(v).reshape(-1))

    def state_dict(self):
        d = {n: self.get_variable(n) for n in self.variable_names()}
        d.update({n + "/Momentum": self.get_variable(n, "Momentum") for n in self.plan.offsets})
        d["main_global_step"] = np.array(self.global_step, dtype=np.int64)     # isprs:1685
        return d

    def load_state_dict(self, d):
        for n in self.variable_names():
            self.set_variable(n, d[n])
        for n in self.plan.offsets:
            if n + "/Momentum" in d:
                self.set_variable(n, d[n + "/Momentum"], "Momentum")
        self.global_step = int(d.get("main_global_step", 0))

    def _touch_f32(self, name):
        """slab `name` was just written as fp32 only (crop / feed, SE and average-pool producers)."""
        if self.ns and name in self.aplanes:
            self.terms_stale.add(name)

    # ------------------------------------------------------------------ views
    def _is_max(self, i):
        q = self.plan.pools[i]
        return q is not None and q[0] == "max"

    def _avg_k(self, i):
        q = self.plan.pools[i]
        return q[1] if q is not None and q[0] == "avg" else 0

    def _stream(self):
        return torch.cuda.current_stream(self.dev).cuda_stream

    def input_slab(self):
        """(tensor, halo, ld) of the conv1 input slab that drs_crop_normalize fills."""
        C, P = self.plan.buffers["x0"]
        return self.x0, P, C

    def _check(self, B, S):
        if B < 1 or S < 1 or B > self.b_max or S > self.s_max:
            raise ValueError("batch %d / patch size %d outside the allocated (%d, %d)" % (B, S, self.b_max, self.s_max))

    # ------------------------------------------------------------------ feeding in the reference's sess.run form
    def feed(self, batch_x, batch_y=None, crop_size=None, mask=None, acc_mask=None):
        """Take the reference's feed_dict: x float32 [B, s*s*C] (row-major NHWC), y [B, s*s] class ids
        (isprs:1746-1747, 1751), optional contest `mask` [B, s*s] (contest:1083-1086)."""
        from .patches import pack_feed
        return pack_feed(self, batch_x, batch_y, crop_size, mask, acc_mask)

    def loss_value(self, loss_parts):
        """total loss = mean CE + sum_k wd * l2_loss(kernel_k) (isprs:1089-1099, 646-651); synchronises."""
        v = loss_parts.detach().cpu().numpy()
        return float(v[0] + self.wd * v[1])

    def get_gradient(self, name):
        off, shape = self.plan.offsets[name]
        return self.grads[off:off + int(np.prod(shape))].detach().cpu().numpy().reshape(shape).copy()
