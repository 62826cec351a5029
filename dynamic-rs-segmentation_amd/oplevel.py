"""The op-level mirror of the step engine: the launch sequence of a forward pass and of a training step spelled out in Python, one
library call per TensorFlow op (include/drs.h, op level).

Host mirror of what one `sess.run` evaluates in the reference (/root/reference/isprs_dilated_random.py): the `_conv_layer` block
:700-723 per layer, the classifier :1024-1031, `loss_def` :1089-1099, `MomentumOptimizer(...).minimize` :1685-1687 and `tf.argmax`
:1690 -- exactly the sequence csrc/engine.hip enqueues.  It is NOT the product path (`DilatedNet(...)` returns engine.EngineNet):
tests/test_gpu_engine.py holds the engine bitwise equal to this class on every net wiring, the per-op tests and tools drive single
entry points through it, and the opt-in split-bf16 arithmetics of the convolutions (csrc/conv_split.hip) exist on this path only.
"""
import torch

from . import _lib
from .net import BN_DECAY, LR_DECAY_STEPS, MOMENTUM, DilatedNet, _ptr


class OpLevelNet(DilatedNet):
    def _alloc_params(self):
        p = self.plan
        f32 = dict(dtype=torch.float32, device=self.dev)
        self.params = torch.zeros(p.n_params, **f32)
        self.grads = torch.zeros(p.n_params, **f32)
        self.mom = torch.zeros(p.n_params, **f32)
        self.bn = torch.zeros(p.n_bn, **f32)

    # ------------------------------------------------------------------ workspaces
    def _alloc(self):
        p, B, S = self.plan, self.b_max, self.s_max
        M = B * S * S
        f32 = dict(dtype=torch.float32, device=self.dev)
        u8 = dict(dtype=torch.uint8, device=self.dev)
        f64 = dict(dtype=torch.float64, device=self.dev)
        L0 = p.layers[0]
        # activation slabs (zero-haloed, see Plan.buffers) and, for every slab but the input, the gradient wrt it [M][C]
        self.abuf = {n: torch.zeros(B * (S + 2 * P) ** 2 * C, **f32) for n, (C, P) in p.buffers.items()}
        self.gbuf = {n: torch.zeros(M * C, **f32) for n, (C, P) in p.buffers.items() if n != "x0"}
        self.x0 = self.abuf["x0"]                     # conv1 input slab (crop target)
        self.z = [torch.zeros(M * L.cout, **f32) for L in p.layers]
        self.idx = [torch.zeros(M * L.cout, **u8) if self._is_max(i) else None for i, L in enumerate(p.layers)]
        self.mean_rstd = [torch.zeros(L.cout * 2, **f32) for L in p.layers]
        cmax = max(L.cout for L in p.layers)
        hmax = max(L.halo for L in p.layers)
        self.sums = torch.zeros(cmax * 2, **f64)
        self.colsum_scratch = torch.zeros(_lib.query("drs_colsum_scratch_doubles", max(2 * cmax, p.c_last * p.K)), **f64)
        rows_fwd = max((M + self._mtile(i) - 1) // self._mtile(i) for i in range(len(p.layers)))
        # the slab's row count depends on the patch size through the kernel's tiling: size it for every S up to s_max
        # (not monotonic in the batch either: every (b, s) a step may be called with)
        part = max(_lib.query("drs_bn_backward_rows", b, s, L.cout, 1 if self._is_max(i) else 0) * L.cout * 2
                   for i, L in enumerate(p.layers) for b in range(1, B + 1) for s in range(1, S + 1))
        self.partial = torch.zeros(max(rows_fwd * cmax * 2, part), **f32)
        if p.se:        # per SE block: the activated input, its spatial mean and the two excitation vectors (kept for backward)
            self.se_state = {}
            for i in p.se:
                C = p.layers[i].cout
                self.se_state[i] = dict(act=torch.zeros(M * C, **f32), s=torch.zeros(B * C, **f32),
                                        e1=torch.zeros(B * (C // 4), **f32), e2=torch.zeros(B * C, **f32))
            self.se_scratch = torch.zeros(B * (3 * cmax + cmax // 4), **f32)
        if p.se or any(q is not None and q[0] == "avg" for q in p.pools):
            self.act = torch.zeros(M * cmax, **f32)        # activated, not yet averaged output of a layer
            self.gpool = torch.zeros(M * cmax, **f32)      # gradient wrt it
        self.gxh = torch.zeros(M * cmax, **f32)
        self.gz = torch.zeros(B * (S + 2 * hmax) ** 2 * cmax, **f32)
        slab = max(_lib.query("drs_conv_wgrad_splits", B, S, L.k, L.cin_k, L.cout) * L.k * L.k * L.cin_k * L.cout
                   for L in p.layers)
        if self.ns:
            # split-bf16 arithmetic: bf16 term planes of every conv input slab and of the haloed output gradient, and the
            # filters in the K-contiguous split form (forward and input-gradient orientation)
            i16 = dict(dtype=torch.int16, device=self.dev)
            ns = self.ns
            srcs = {L.src for i, L in enumerate(p.layers) if self._split_fwd(i)}
            self.aplanes = {n: torch.zeros(ns * self.abuf[n].numel(), **i16) for n in srcs}
            # slabs somebody still reads as fp32 (the classifier, convolutions that stay on the fp32 kernels); the other
            # slabs exist as bf16 terms only, written by the producing kernel itself
            self.f32_slabs = {p.feat} | {L.src for i, L in enumerate(p.layers) if not self._split_fwd(i)} | (set(p.buffers) - srcs)
            self.terms_stale = set(self.aplanes)
            self.gzplanes = torch.zeros(ns * self.gz.numel(), **i16)
            self.wf_planes = [torch.zeros(ns * L.k * L.k * L.cin_k * L.cout, **i16) if self._split_fwd(i) else None
                       for i, L in enumerate(p.layers)]
            self.wd_planes = [torch.zeros(ns * L.k * L.k * L.cin * L.cout, **i16) if self._split_dgrad(i) else None
                       for i, L in enumerate(p.layers)]
            slab = max([slab] + [_lib.query("drs_conv_wgrad_split_splits", B, S, L.k, L.cin_k, L.cout, L.halo, self.ns) * L.k * L.k * L.cin_k * L.cout
                                 for i, L in enumerate(p.layers) if self._split_fwd(i)])
        self.slab = torch.zeros(slab, **f32)
        # partial-sum slab of the stream-K convolution launches (forward: N = cout; input gradient: N = cin)
        self.conv_ws = torch.zeros(max(1, max(max(_lib.query("drs_conv_workspace_floats", L.cout), _lib.query("drs_conv_workspace_floats", L.cin) if i else 0)
                                              for i, L in enumerate(p.layers))), **f32)
        # conv1's filter with the bands padded to cin_k; with cin_k < 32 its rows are padded (zeros) to a whole number of K-steps
        self.w0pad = torch.zeros(-(-L0.k * L0.k * L0.cin_k // 32) * 32 * L0.cout, **f32)
        self.wt = [None] + [torch.zeros(L.k * L.k * L.cin * L.cout, **f32) for L in p.layers[1:]]
        crow = _lib.query("drs_classifier_rows", B, S)
        self.dw_partial = torch.zeros(crow * p.c_last * p.K, **f32)
        self.db_partial = torch.zeros(crow * p.K, **f32)
        self.loss_partial = torch.zeros(crow, **f64)
        self.scalars = torch.zeros(4, **f64)          # [0] CE sum (this rank), [1] l2, [2..3] spare
        self.l2_scratch = torch.zeros(256, **f64)
        self.logits = torch.zeros(M * p.K, **f32)
        self.pred = torch.zeros(M, **u8)
        self.conf = torch.zeros(p.K * p.K, dtype=torch.int32, device=self.dev)
        self.labels = torch.zeros(M, **u8)
        self.acc_mask = torch.ones(M, **u8)
        self.loss_mask = torch.ones(M, **u8)

    def workspace_bytes(self):
        tot = 0
        for v in vars(self).values():
            for t in (v if isinstance(v, list) else [v]):
                if isinstance(t, torch.Tensor):
                    tot += t.numel() * t.element_size()
        return tot

    # ------------------------------------------------------------------ split-bf16 arithmetic
    def _split_fwd(self, i):
        """conv block i runs its forward and filter-gradient passes on the split-bf16 kernels (tile shapes need Cout % 64)."""
        L = self.plan.layers[i]
        return self.ns > 0 and L.cout % 64 == 0 and L.cin_k % 32 == 0

    def _mtile(self, i):
        """pixels per row of the batch-norm statistics slab the forward convolution of block i writes"""
        return _lib.query("drs_split_conv_mtile" if self._split_fwd(i) else "drs_conv_mtile", self.plan.layers[i].cout)

    def _split_dgrad(self, i):
        """... and its input-gradient pass (a GEMM with N = Cin)."""
        L = self.plan.layers[i]
        return self._split_fwd(i) and L.src != "x0" and L.cin % 64 == 0

    def _split_slab(self, name, B, S):
        """bf16 terms of activation slab `name` from its fp32 image, unless every write of this pass already came with terms."""
        if name not in self.terms_stale:
            return
        C, P = self.plan.buffers[name]
        n = B * (S + 2 * P) ** 2 * C
        t = self.abuf[name]
        self._k("split", n * (4.0 + 2.0 * self.ns), "drs_split_terms", _ptr(t), n, self.ns, _ptr(self.aplanes[name]), self._stream())
        self.terms_stale.discard(name)

    def _in_view(self, i):
        """(tensor, halo, ld, coff) of the input of conv block i: channels [0, cin) of its source slab."""
        L = self.plan.layers[i]
        C, P = self.plan.buffers[L.src]
        return self.abuf[L.src], P, C, 0

    def _out_view(self, i):
        """where the activated (pooled) output of conv block i goes: a channel slice of its destination slab."""
        L = self.plan.layers[i]
        C, P = self.plan.buffers[L.dst]
        return self.abuf[L.dst], P, C, L.dst_coff

    def _feat_view(self):
        C, P = self.plan.buffers[self.plan.feat]
        return self.abuf[self.plan.feat], P, C, 0

    def _k(self, kind, work, name, *args):
        """enqueue one library call; with a KernelTimer attached, bracket it with HIP events."""
        if self.timer is None:
            return _lib.call(name, *args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.call(name, *args)
        e1.record()
        self.timer.add(kind, work, e0, e1)

    # ------------------------------------------------------------------ forward
    def _prepare_weights(self, st, training):
        p = self.plan
        L0 = p.layers[0]
        off, _ = p.offsets[L0.name + "/weights"]
        _lib.call("drs_filter_pad_cin", self.params[off:].data_ptr(), _ptr(self.w0pad), L0.k, L0.cin, L0.cin_k, L0.cout, st)
        for i, L in enumerate(p.layers):
            if self._split_fwd(i):
                off, _ = p.offsets[L.name + "/weights"]
                _lib.call("drs_filter_split", self.params[off:].data_ptr(), L.k, L.cin, L.cin_k, L.cout, self.ns, _ptr(self.wf_planes[i]),
                          _ptr(self.wd_planes[i]) if training else None, st)

    def _weight_ptr(self, i):
        if i == 0:
            return self.w0pad.data_ptr()
        off, _ = self.plan.offsets[self.plan.layers[i].name + "/weights"]
        return self.params[off:].data_ptr()

    def _pptr(self, name, flat=None):
        off, _ = self.plan.offsets[name]
        return (self.params if flat is None else flat)[off:].data_ptr()

    def _bias_ptr(self, name):
        off, _ = self.plan.offsets[name + "/biases"]
        return self.params[off:].data_ptr()

    def _forward_layers(self, B, S, training, count):
        p, st = self.plan, self._stream()
        M = B * S * S
        self._prepare_weights(st, training)
        self._touch_f32("x0")
        halo_ok = self.__dict__.setdefault("_halo_zeroed", {})     # slab -> (B, S) of the pooling call that last zeroed its halo
        for i, L in enumerate(p.layers):
            xin, Pin, ldin, cin_off = self._in_view(i)
            stats = self.partial if training else None
            if self._split_fwd(i):
                self._split_slab(L.src, B, S)
                self._k("conv_fwd", 2.0 * M * L.k * L.k * L.cin * L.cout, "drs_conv_forward_split", _ptr(self.aplanes[L.src]),
                        B, S, Pin, ldin, cin_off, _ptr(self.wf_planes[i]), self._bias_ptr(L.name),
                        L.k, L.rate, L.pad_b, L.cin_k, L.cout, _ptr(self.z[i]), L.cout, 0, 0, _ptr(stats), self.ns, st)
            else:
                self._k("conv_fwd", 2.0 * M * L.k * L.k * L.cin * L.cout, "drs_conv_forward_ws", _ptr(xin), B, S, Pin, ldin, cin_off,
                        self._weight_ptr(i), self._bias_ptr(L.name),
                        L.k, L.rate, L.pad_b, L.cin_k, L.cout, _ptr(self.z[i]), L.cout, 0, 0, _ptr(stats), _ptr(self.conv_ws), self.conv_ws.numel(), st)
            bo = p.bn_offsets[L.name]
            mm, mv = self.bn[bo:bo + L.cout], self.bn[bo + L.cout:bo + 2 * L.cout]
            if training and not self.comm.collective:
                # tile statistics -> (mean, rstd) and the moving averages in one launch
                _lib.call("drs_conv_stats_finish", _ptr(self.partial), M, self._mtile(i), L.cout, float(count), _ptr(self.mean_rstd[i]),
                          _ptr(mm), _ptr(mv), BN_DECAY, self.bessel, None, st)
            elif training:
                _lib.call("drs_conv_stats_reduce", _ptr(self.partial), M, self._mtile(i), L.cout, _ptr(self.sums), None, st)
                self.comm.all_reduce_sum(self.sums[:2 * L.cout])           # sync batch norm over the global batch
                _lib.call("drs_bn_finish", _ptr(self.sums), float(count), L.cout, _ptr(self.mean_rstd[i]), _ptr(mm), _ptr(mv),
                          BN_DECAY, self.bessel, st)
            else:
                _lib.call("drs_bn_eval_coeffs", _ptr(mm), _ptr(mv), L.cout, _ptr(self.mean_rstd[i]), st)
            out, Pout, ldout, coff = self._out_view(i)
            mx, ak = self._is_max(i), self._avg_k(i)
            if i in p.se:   # activation into a plain [M][C] buffer, then squeeze-and-excitation scaling into the next slab
                stt, sc = self.se_state[i], p.se[i]
                self._k("bn_act_pool_fwd", M * L.cout * 8.0, "drs_bn_act_pool_forward", _ptr(self.z[i]), B, S, L.cout,
                        _ptr(self.mean_rstd[i]), p.alpha, 0, _ptr(stt["act"]), 0, L.cout, 0, None, st)
                self._k("se_fwd", M * L.cout * 12.0, "drs_se_forward", _ptr(stt["act"]), B, S, L.cout, L.cout // 4,
                        self._pptr(sc + "_fc1/weights"), self._pptr(sc + "_fc1/biases"), self._pptr(sc + "_fc2/weights"),
                        self._pptr(sc + "_fc2/biases"), _ptr(stt["s"]), _ptr(stt["e1"]), _ptr(stt["e2"]), _ptr(out), Pout, ldout, coff, st)
                self._touch_f32(L.dst)
            elif ak:    # activation into a plain [M][C] buffer, then the k x k average into the next layer's slab
                self._k("bn_act_pool_fwd", M * L.cout * 8.0, "drs_bn_act_pool_forward", _ptr(self.z[i]), B, S, L.cout,
                        _ptr(self.mean_rstd[i]), p.alpha, 0, _ptr(self.act), 0, L.cout, 0, None, st)
                self._k("avg_pool_fwd", M * L.cout * 8.0, "drs_avg_pool_forward", _ptr(self.act), B, S, L.cout, ak, _ptr(out), Pout,
                        ldout, coff, st)
                self._touch_f32(L.dst)
            elif self.ns and L.dst in self.aplanes and ldout == L.cout and coff == 0:
                # the producer of a whole slab writes the bf16 terms the next convolution reads itself (and the fp32 image only
                # if someone needs it); channel slices of a shared slab (dense / squeeze nets) go through drs_split_terms
                keep = L.dst in self.f32_slabs
                self.terms_stale.discard(L.dst)
                hz = 2 if (mx and halo_ok.get(L.dst) == (B, S, keep)) else 0
                halo_ok[L.dst] = (B, S, keep) if mx else None
                self._k("bn_act_pool_fwd", M * L.cout * ((5.0 if (training and mx) else 4.0) + 2.0 * self.ns + (4.0 if keep else 0.0)),
                        "drs_bn_act_pool_forward_terms", _ptr(self.z[i]), B, S, L.cout, _ptr(self.mean_rstd[i]), p.alpha,
                        (1 if mx else 0) | hz, _ptr(out) if keep else None, Pout, ldout, coff,
                        _ptr(self.idx[i]) if (training and mx) else None, _ptr(self.aplanes[L.dst]), self.ns, st)
            else:
                # the halo of a slab this block owns alone stays zero between calls of the same geometry: do not rewrite it
                whole = ldout == L.cout and coff == 0
                hz = 2 if (mx and whole and halo_ok.get(L.dst) == (B, S, True)) else 0
                halo_ok[L.dst] = (B, S, True) if (mx and whole) else None
                self._k("bn_act_pool_fwd", M * L.cout * (9.0 if (training and mx) else 8.0), "drs_bn_act_pool_forward",
                        _ptr(self.z[i]), B, S, L.cout, _ptr(self.mean_rstd[i]), p.alpha, (1 if mx else 0) | hz, _ptr(out), Pout, ldout,
                        coff, _ptr(self.idx[i]) if (training and mx) else None, st)
                self._touch_f32(L.dst)

    def forward(self, B, S, want_logits=True, labels=False, acc_mask=False, ignore_label=-1):
        """is_training=False pass over the slab filled by crop/feed: returns (pred uint8 [B,S,S] device,
        logits float32 [B,S,S,K] device or None).  With labels=True the confusion matrix of (self.labels,
        pred) is added into self.conf (validation, isprs:1599)."""
        self._check(B, S)
        p, st = self.plan, self._stream()
        self._forward_layers(B, S, False, B * S * S)
        feat, Pf, ldf, cf = self._feat_view()
        off, _ = p.offsets["conv_classifier/weights"]
        _lib.call("drs_classifier_loss", _ptr(feat), B, S, Pf, ldf, cf, p.c_last, p.K, self.params[off:].data_ptr(),
                  self._bias_ptr("conv_classifier"), None, None, None, 0.0, _ptr(self.logits) if want_logits else None,
                  _ptr(self.pred), None, 0, 0, None, None, None, None, st)
        M = B * S * S
        if labels:
            _lib.call("drs_confusion", _ptr(self.labels), _ptr(self.pred), _ptr(self.acc_mask) if acc_mask else None, M, p.K,
                      ignore_label, _ptr(self.conf), st)
        return self.pred[:M].view(B, S, S), (self.logits[:M * p.K].view(B, S, S, p.K) if want_logits else None)

    # ------------------------------------------------------------------ training step
    def learning_rate(self, lr0):
        """tf.train.exponential_decay(lr0, global_step, 50000, factor, staircase=True) (isprs:1686)."""
        return lr0 * self.lr_decay_factor ** (self.global_step // LR_DECAY_STEPS)

    def train_step(self, B, S, lr0, use_loss_mask=False, use_acc_mask=True, global_pixels=None, apply_update=True,
                   want_logits=False):
        """One optimisation step on the slab / labels / masks currently on the device.
        Returns a dict of DEVICE tensors (no host synchronisation):
          loss_parts  float64 [2] = (sum of CE over this rank's pixels / N_global after all-reduce, 0.5*sum w^2)
          pred        uint8 [B,S,S];  conf int32 [K,K] (this step, this rank's pixels, all-reduced)
        `global_pixels` = number of pixels the loss averages over on ALL ranks (defaults to B*S*S*world; the
        contest form passes the number of unmasked pixels).  Every rank must hold the same B (the batch-norm
        count is B*S*S*world)."""
        self._check(B, S)
        p, st = self.plan, self._stream()
        M = B * S * S
        n_bn = float(M * self.comm.world)        # batch-norm statistics run over every pixel of the global batch
        # (<= 0, like None, means "every pixel of every rank": the step engine's rule, csrc/engine.hip train_step_impl -- a mask that
        #  leaves no pixel must not divide by zero here and by B*S*S there)
        n_glob = float(global_pixels if global_pixels is not None and global_pixels > 0 else M * self.comm.world)
        self._forward_layers(B, S, True, n_bn)
        nL = len(p.layers)
        for i in range(1, nL):
            L = p.layers[i]
            if self._split_dgrad(i):
                continue            # the split path's input-gradient filter was written by drs_filter_split
            off, _ = p.offsets[L.name + "/weights"]
            _lib.call("drs_filter_flip_transpose", self.params[off:].data_ptr(), _ptr(self.wt[i]), L.k, L.cin, L.cout, st)
        # classifier + loss + gradient wrt the features
        feat, Pf, ldf, cf = self._feat_view()
        woff, _ = p.offsets["conv_classifier/weights"]
        boff, _ = p.offsets["conv_classifier/biases"]
        gfeat, ldg, cg = self.gbuf[p.feat], p.buffers[p.feat][0], 0
        self.conf.zero_()
        self._k("classifier_loss", M * p.c_last * 8.0, "drs_classifier_loss", _ptr(feat), B, S, Pf, ldf, cf, p.c_last, p.K,
                self.params[woff:].data_ptr(), self.params[boff:].data_ptr(), _ptr(self.labels), _ptr(self.loss_mask) if use_loss_mask else None,
                  _ptr(self.acc_mask) if use_acc_mask else None, 1.0 / n_glob, _ptr(self.logits) if want_logits else None,
                  _ptr(self.pred), _ptr(gfeat), ldg, cg, _ptr(self.dw_partial), _ptr(self.db_partial), _ptr(self.loss_partial),
                  _ptr(self.conf), st)
        crow = _lib.query("drs_classifier_rows", B, S)
        _lib.call("drs_rows_reduce_f32", _ptr(self.dw_partial), crow, p.c_last * p.K, self.grads[woff:].data_ptr(), _ptr(self.colsum_scratch), st)
        _lib.call("drs_rows_reduce_f32", _ptr(self.db_partial), crow, p.K, self.grads[boff:].data_ptr(), _ptr(self.colsum_scratch), st)
        _lib.call("drs_sum_f64", _ptr(self.loss_partial), crow, _ptr(self.scalars), st)
        _lib.call("drs_l2_loss", _ptr(self.params), p.n_decay, _ptr(self.l2_scratch), self.scalars[1:].data_ptr(), st)
        # conv biases sit in front of a mean-subtracting batch norm: their gradient is identically zero
        b0, _ = p.offsets[p.layers[0].name + "/biases"]
        self.grads[b0:boff].zero_()
        # gradient all-reduce in buckets that overlap the rest of the backward pass (collectives run on RCCL's own
        # stream): kernel gradients go as their layers finish, last layers first (they hold most of the bytes: conv7+conv8
        # = 49 % of Dilated8Pooling); the small classifier / SE / bias tail goes last
        pending = []
        bucket_hi = woff                       # kernels [bucket_lo, bucket_hi) of the flat buffer are still to be sent
        # reverse loop over the conv blocks.  The gradient wrt a slab is first SET (by the classifier, or by the first block
        # that propagates into it) and then ACCUMULATED into by every further reader of that slab (dense / squeeze nets)
        written = {p.feat}

        def filter_gradient(i):
            """filter gradient of conv block i from its input slab and the gz currently in place; then, under data parallelism,
            the gradient bucket that this layer completes"""
            nonlocal bucket_hi
            L = p.layers[i]
            xin, Pin, ldin, cin_off = self._in_view(i)
            goff, _ = p.offsets[L.name + "/weights"]
            if self._split_fwd(i):
                self._k("conv_wgrad", 2.0 * M * L.k * L.k * L.cin * L.cout, "drs_conv_wgrad_split", _ptr(self.aplanes[L.src]),
                        B, S, Pin, ldin, cin_off, _ptr(self.gzplanes), L.halo, L.cout, 0, L.k, L.rate,
                        L.pad_b, L.cin_k, L.cin, L.cout, _ptr(self.slab), self.grads[goff:].data_ptr(), self.ns, st)
            else:
                self._k("conv_wgrad", 2.0 * M * L.k * L.k * L.cin * L.cout, "drs_conv_wgrad", _ptr(xin), B, S, Pin, ldin, cin_off,
                        _ptr(self.gz), L.halo, L.cout, 0, L.k, L.rate, L.pad_b, L.cin_k, L.cin, L.cout, _ptr(self.slab),
                        self.grads[goff:].data_ptr(), st)
            if self.comm.collective and i > 0 and (nL - i) % 2 == 0:                   # every second layer: one bucket
                pending.append(self.comm.all_reduce_sum_async(self.grads[goff:bucket_hi]))
                bucket_hi = goff

        deferred_wgrad = None
        for i in reversed(range(nL)):
            L = p.layers[i]
            gcur, ldc, cc = self.gbuf[L.dst], p.buffers[L.dst][0], L.dst_coff
            mx, ak = self._is_max(i), self._avg_k(i)
            if i in p.se:
                stt, sc = self.se_state[i], p.se[i]
                self._k("se_bwd", M * L.cout * 16.0, "drs_se_backward", _ptr(gcur), ldc, cc, _ptr(stt["act"]), _ptr(stt["s"]),
                        _ptr(stt["e1"]), _ptr(stt["e2"]), self._pptr(sc + "_fc1/weights"), self._pptr(sc + "_fc2/weights"), B, S, L.cout,
                        L.cout // 4, _ptr(self.gpool), self._pptr(sc + "_fc1/weights", self.grads), self._pptr(sc + "_fc1/biases", self.grads),
                        self._pptr(sc + "_fc2/weights", self.grads), self._pptr(sc + "_fc2/biases", self.grads), _ptr(self.se_scratch), st)
                gsrc, lds_, cs_ = self.gpool, L.cout, 0
            elif ak:
                self._k("avg_pool_bwd", M * L.cout * 8.0, "drs_avg_pool_backward", _ptr(gcur), ldc, cc, B, S, L.cout, ak,
                        _ptr(self.gpool), st)
                gsrc, lds_, cs_ = self.gpool, L.cout, 0
            else:
                gsrc, lds_, cs_ = gcur, ldc, cc
            self._k("bn_bwd_reduce", M * L.cout * (13.0 if mx else 12.0), "drs_bn_backward_reduce", _ptr(gsrc), lds_, cs_,
                    _ptr(self.z[i]), _ptr(self.idx[i]), B, S, L.cout, _ptr(self.mean_rstd[i]), p.alpha, 1 if mx else 0,
                    _ptr(self.gxh), _ptr(self.partial), st)
            _lib.call("drs_stats_reduce", _ptr(self.partial), _lib.query("drs_bn_backward_rows", B, S, L.cout, 1 if mx else 0), L.cout,
                      _ptr(self.sums), _ptr(self.colsum_scratch), st)
            # sync batch norm: the all-reduce of (sum g, sum g*xhat) runs on the collective's stream while this stream computes the
            # filter gradient of the block above (deferred to here: it only needs that block's gz, which is still in place)
            h_bn = self.comm.all_reduce_sum_async(self.sums[:2 * L.cout])
            if deferred_wgrad is not None:
                deferred_wgrad()
            self.comm.wait([h_bn])
            if self._split_fwd(i):
                keep = self.debug is not None or (L.src != "x0" and not self._split_dgrad(i))     # an fp32 kernel still reads gz
                self._k("bn_bwd_apply", M * L.cout * (8.0 + 2.0 * self.ns + (4.0 if keep else 0.0)), "drs_bn_backward_apply_terms",
                        _ptr(self.gxh), _ptr(self.z[i]), B, S, L.cout, _ptr(self.mean_rstd[i]), _ptr(self.sums), n_bn,
                        _ptr(self.gz) if keep else None, L.halo, L.cout, 0, _ptr(self.gzplanes), self.ns, st)
            else:
                self._k("bn_bwd_apply", M * L.cout * 12.0, "drs_bn_backward_apply", _ptr(self.gxh), _ptr(self.z[i]), B, S, L.cout,
                        _ptr(self.mean_rstd[i]), _ptr(self.sums), n_bn, _ptr(self.gz), L.halo, L.cout, 0, st)
            if self.debug is not None:      # diagnostics only: per-layer snapshots for tests/diag_net.py
                self.debug["gxh%d" % i] = self.gxh[:M * L.cout].clone()
                self.debug["gz%d" % i] = self.gz[:B * (S + 2 * L.halo) ** 2 * L.cout].clone()
            if L.src != "x0":
                acc = 1 if L.src in written else 0
                written.add(L.src)
                if self._split_dgrad(i):
                    self._k("conv_dgrad", 2.0 * M * L.k * L.k * L.cin * L.cout, "drs_conv_forward_split", _ptr(self.gzplanes),
                            B, S, L.halo, L.cout, 0, _ptr(self.wd_planes[i]), None, L.k, L.rate,
                            L.pad_a, L.cout, L.cin, _ptr(self.gbuf[L.src]), p.buffers[L.src][0], 0, acc, None, self.ns, st)
                else:
                    self._k("conv_dgrad", 2.0 * M * L.k * L.k * L.cin * L.cout, "drs_conv_forward_ws", _ptr(self.gz), B, S, L.halo, L.cout,
                            0, _ptr(self.wt[i]), None, L.k, L.rate, L.pad_a, L.cout, L.cin, _ptr(self.gbuf[L.src]), p.buffers[L.src][0], 0,
                            acc, None, _ptr(self.conv_ws), self.conv_ws.numel(), st)
            deferred_wgrad = (lambda i=i: filter_gradient(i))
        deferred_wgrad()
        if self.comm.collective:
            pending.append(self.comm.all_reduce_sum_async(self.grads[0:bucket_hi]))      # the remaining (earliest) layers
            pending.append(self.comm.all_reduce_sum_async(self.grads[woff:]))            # classifier, SE layers and every bias (small)
            self.comm.all_reduce_sum(self.scalars[:1])
            self.comm.all_reduce_sum(self.conf)
            self.comm.wait(pending)
        self.scalars[0:1].mul_(1.0 / n_glob)
        if apply_update:
            self.apply_update(lr0)
        return dict(loss_parts=self.scalars[:2], pred=self.pred[:M].view(B, S, S), conf=self.conf.view(p.K, p.K))

    def apply_update(self, lr0):
        p, st = self.plan, self._stream()
        self._k("momentum_update", p.n_params * 20.0, "drs_momentum_update", _ptr(self.params), _ptr(self.grads), _ptr(self.mom),
                p.n_params, p.n_decay, self.learning_rate(lr0), self.wd, MOMENTUM, 1.0, st)
        self.global_step += 1
