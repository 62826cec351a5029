"""DilatedNet on the step-level C ABI (include/drs.h, csrc/engine.hip): one library call per `sess.run`.

The net table, the variable layout and the launch order of a step live in the library (`drs_net_create`, `drs_train_step`,
`drs_forward`); this class allocates the buffers the library lists (PyTorch-ROCm tensors: device memory and the stream are all
PyTorch supplies), binds them, and hands the library an all-reduce callback for data parallelism (torch.distributed over RCCL).
Everything else -- feeding, variable access by TensorFlow scope name, checkpoints, the loops -- is inherited from the op-level
mirror `net.DilatedNet`, which keeps the per-op launch sequence in Python and is held bitwise equal to this class by
tests/test_gpu_engine.py.
"""
import ctypes as C

import torch

from . import _lib
from .net import DilatedNet, KernelTimer

_DTYPES = {0: torch.float32, 1: torch.float64, 2: torch.uint8, 3: torch.int32}


RCCL_FORM_LABELS = {1: "rccl (inline: one communicator on the compute stream)", 2: "rccl (asynchronous: DRS_RCCL_ASYNC)",
                    3: "rccl (inline + two overlapped gradient buckets: DRS_RCCL_BUCKETS)"}


def call_with_timeout(fn, seconds, what):
    """run fn() in a helper thread; its result, or its exception re-raised here, or DrsError after `seconds` without a return (the
    thread is left behind as a daemon: the call it is stuck in cannot be cancelled)"""
    import threading
    box = {}

    def run():
        try:
            box["value"] = fn()
        except BaseException as e:       # handed to the caller's thread
            box["error"] = e
    t = threading.Thread(target=run, daemon=True, name="drs-timeout-call")
    t.start()
    t.join(seconds)
    if t.is_alive():
        raise _lib.DrsError("%s did not return within %.0f s" % (what, seconds))
    if "error" in box:
        raise box["error"]
    return box.get("value")


class EngineTimer(KernelTimer):
    """bench.py's per-kernel-family figures, measured by the library (HIP events around its launches)."""

    def __init__(self, net):
        KernelTimer.__init__(self)
        self.net = net

    def summary(self):
        out = {}
        name = C.create_string_buffer(64)
        n, ms, work = C.c_int(), C.c_double(), C.c_double()
        for k in range(_lib.query("drs_net_num_timing_kinds")):
            _lib.call("drs_net_timing_summary", self.net.h, k, name, 64, C.byref(n), C.byref(ms), C.byref(work))
            if n.value:
                out[name.value.decode()] = dict(launches=n.value, ms=ms.value, work=work.value)
        return out


class EngineNet(DilatedNet):
    h = None
    ranks_observed = 1          # ranks a sum of ones over the step's communicator came back with (set when a communicator is installed)

    # ------------------------------------------------------------------ buffers
    def _alloc_params(self):
        """create the library-side net and every buffer it lists; the flat parameter / gradient / momentum / statistics buffers are
        among them"""
        if self.ns:
            raise ValueError("the step-level library runs the exact-fp32 arithmetic; split-bf16 is the op-level path (engine=False)")
        hp = C.c_void_p()
        _lib.call("drs_net_create", self.plan.net_type.encode(), self.plan.channels, self.plan.K, self.wd, self.b_max, self.s_max, self.bessel,
                  float(self.lr_decay_factor), C.byref(hp))
        self.h = hp
        self._bufs = {}
        name, nb, dt = C.create_string_buffer(64), C.c_size_t(), C.c_int()
        for i in range(_lib.query("drs_net_num_buffers", self.h)):
            _lib.call("drs_net_buffer_info", self.h, i, name, 64, C.byref(nb), C.byref(dt))
            dtype = _DTYPES[dt.value]
            t = torch.zeros(nb.value // torch.empty(0, dtype=dtype).element_size(), dtype=dtype, device=self.dev)
            self._bufs[name.value.decode()] = t
            _lib.call("drs_net_bind", self.h, name.value, t.data_ptr(), nb.value)
        b = self._bufs
        self.params, self.grads, self.mom, self.bn = b["params"], b["grads"], b["momentum"], b["bn"]
        # the library's layout must be the one nets.Plan describes (names, offsets): both are derived from the same tables
        np_, nd, nbn, nl, c0, p0 = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_int(), C.c_int(), C.c_int()
        _lib.call("drs_net_layout", self.h, C.byref(np_), C.byref(nd), C.byref(nbn), C.byref(nl), C.byref(c0), C.byref(p0))
        p = self.plan
        assert (np_.value, nd.value, nbn.value, nl.value) == (p.n_params, p.n_decay, p.n_bn, len(p.layers)), "library / nets.Plan layout mismatch"
        assert (c0.value, p0.value) == p.buffers["x0"]
        self._gs_pending = getattr(self, "_gs_pending", 0)
        _lib.query("drs_net_global_step", self.h, int(self._gs_pending))
        if self.comm.collective:
            self._install_comm()

    def _alloc(self):
        """aliases of the library's buffers under the op-level attribute names (tests, feeding and the loops read them)"""
        b, p = self._bufs, self.plan
        self.abuf = {n: b["act:" + n] for n in p.buffers}
        self.gbuf = {n: b["gact:" + n] for n in p.buffers if n != "x0"}
        self.x0 = self.abuf["x0"]
        nl = len(p.layers)
        self.z = [b["z%d" % i] for i in range(nl)]
        self.idx = [b.get("idx%d" % i) for i in range(nl)]
        self.mean_rstd = [b["mean_rstd%d" % i] for i in range(nl)]
        for n in ("sums", "partial", "gxh", "gz", "slab", "w0pad", "dw_partial", "db_partial", "loss_partial", "scalars", "logits", "pred", "conf",
                  "labels", "acc_mask", "loss_mask"):
            setattr(self, n, b[n])
        self.acc_mask.fill_(1)
        self.loss_mask.fill_(1)

    def workspace_bytes(self):
        return sum(t.numel() * t.element_size() for t in self._bufs.values())

    def close(self):
        """destroy the library's net and, after it, the RCCL communicators it was given (a data-parallel program calls this before it
        tears its process group down; __del__ does the same when the object goes away)"""
        if self.h is not None and _lib._lib is not None:
            _lib.load().drs_net_destroy(self.h)
            self.h = None
            for c in getattr(self, "_rccl", []):
                _lib.load().drs_rccl_comm_destroy(c)
            self._rccl = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ data parallelism: the library's all-reduce callback
    def _install_comm(self):
        """data parallelism: the step's sums are issued by the library itself through RCCL (drs_net_set_rccl) when the process group
        is RCCL ('nccl') -- the host only carries the two 128-byte communicator ids from rank 0 to the other ranks; otherwise (gloo
        rehearsals and CPU tests, DRS_COMM=torch, or RCCL not bindable) through the all-reduce callback into torch.distributed."""
        import os
        # (DRS_COMM=rccl over a host group that is not RCCL -- gloo -- still takes the library-side path: the host group only carries
        #  the communicator ids.  The tests run the world > 1 code of the library-side collectives that way on a one-GPU box, after
        #  naming a shared-memory stand-in for RCCL through drs_rccl_bind_library; tests/test_gpu_dp.py.)
        forced = os.environ.get("DRS_COMM") == "rccl"
        if (getattr(self.comm, "backend", None) == "nccl" or forced) and os.environ.get("DRS_COMM", "rccl") != "torch":
            err = None
            if self.comm.all_true(bool(_lib.query("drs_rccl_available"))):
                try:
                    self._install_rccl()
                except Exception as e:
                    err = e
            else:
                err = _lib.DrsError("librccl could not be bound on every rank")
            # the choice of path is itself collective: every rank takes the callback unless every rank holds two working communicators
            ok = self.comm.all_true(err is None)
            if ok:
                # the form is the LIBRARY's reading of the environment (drs_rccl_form: one parser); the label says what really runs
                form = getattr(self, "_rccl_form", 1 if len(self._rccl) == 1 else 2)
                single = len(self._rccl) == 1
                _lib.call("drs_net_set_rccl", self.h, self.comm.world, self.comm.rank, self._rccl[0], None if single else self._rccl[1], None)
                self.collectives = RCCL_FORM_LABELS[1 if single else form]
                return
            if os.environ.get("DRS_COMM") == "rccl":
                raise err or _lib.DrsError("library-side RCCL collectives failed on another rank")
            if self.comm.rank == 0 or err is not None:
                print("drs: library-side RCCL collectives unavailable (%r on rank %d); using the torch.distributed callback" % (err, self.comm.rank))
        self._install_callback()

    def _install_rccl(self):
        import os
        # one communicator, driven from the compute stream alone (the library's default, inline form); DRS_RCCL_ASYNC=1 or
        # DRS_RCCL_BUCKETS=2: a small (latency-bound sums) and a big (gradient buckets) one.  The library parses the variables
        # (drs_rccl_form); this side only asks, so that host and library cannot read them differently.
        self._rccl_form = int(_lib.query("drs_rccl_form"))
        ncomm = 1 if self._rccl_form == 1 else 2
        ok, payload = True, []
        if self.comm.rank == 0:
            try:
                for _ in range(ncomm):
                    buf = (C.c_ubyte * 128)()
                    _lib.call("drs_rccl_unique_id", buf)
                    payload.append(bytes(buf))
            except Exception as e:          # every rank must leave the id exchange together (loops.rank0_call does the same)
                ok, payload = False, repr(e)
        ok, payload = self.comm.broadcast_object((ok, payload), src=0)
        if not ok:
            raise _lib.DrsError("rank 0 could not make the RCCL ids: %s" % payload)
        ids = payload
        torch.cuda.set_device(self.dev)
        self._rccl = []
        # ncclCommInitRank is collective: a rank that fails BEFORE it gets there (or inside it) leaves the others waiting in it for good,
        # and nothing above this call would ever run again (outside bench.py no watchdog exists).  So the call runs in a helper thread
        # under a rank-local limit; a rank whose call does not return in time raises here, reaches the collective "did it work
        # everywhere?" question of _install_comm with a no, and every rank takes the callback path together.  (The thread that is
        # still inside RCCL is a daemon: it goes with the process.)
        limit = float(os.environ.get("DRS_RCCL_INIT_TIMEOUT_S", "90"))
        import time
        # first contact with RCCL at this world size, for the record (bench.py: extra.first_contact): seconds inside ncclCommInitRank per
        # communicator, seconds of the first all-reduce (connection set-up + kernel load happen there)
        self.first_contact = dict(comm_init_s=[], first_allreduce_s=None)
        create_err = None
        abandoned = [False]       # set when this rank gives up: a create() still inside RCCL destroys what it gets, late, instead of leaking it
        try:
            for raw in ids:
                h = C.c_void_p()
                def create(raw=raw, h=h):
                    torch.cuda.set_device(self.dev)      # (the current device is per thread)
                    _lib.call("drs_rccl_comm_create", self.comm.world, self.comm.rank, (C.c_ubyte * 128).from_buffer_copy(raw), C.byref(h))
                    if abandoned[0] and h.value:
                        _lib.load().drs_rccl_comm_destroy(h)
                        h.value = None
                t0 = time.perf_counter()
                call_with_timeout(create, limit, "drs_rccl_comm_create (ncclCommInitRank, world %d, rank %d)" % (self.comm.world, self.comm.rank))
                self.first_contact["comm_init_s"].append(round(time.perf_counter() - t0, 3))
                self._rccl.append(h)
        except Exception as e:
            create_err = e
            abandoned[0] = True
        # "did every rank get its communicators?" over the HOST group, before anything is issued on the new ones: a rank whose create
        # timed out or raised never enters the known-answer sums below, and the others would wait in them without a limit
        if not self.comm.all_true(create_err is None):
            for h in self._rccl:
                try:
                    _lib.load().drs_rccl_comm_destroy(h)
                except Exception:
                    pass
            self._rccl = []
            raise create_err or _lib.DrsError("another rank could not create its RCCL communicators")
        # known-answer check of both communicators through the call the step engine issues, in each of its three types
        W, r = self.comm.world, self.comm.rank
        st = torch.cuda.current_stream(self.dev).cuda_stream
        bad = []
        for h in self._rccl:                   # (every rank issues all six, whatever it finds: the calls are collective)
            for dt, code in ((torch.float32, 0), (torch.float64, 1), (torch.int32, 3)):
                t = torch.tensor([r + 1, 1, -(r + 1) * 3], dtype=dt, device=self.dev)
                t0 = time.perf_counter()
                _lib.call("drs_rccl_all_reduce", h, t.data_ptr(), 3, code, st)
                got = t.cpu().tolist()
                if self.first_contact["first_allreduce_s"] is None:
                    self.first_contact["first_allreduce_s"] = round(time.perf_counter() - t0, 3)
                if got != [W * (W + 1) // 2, W, -3 * (W * (W + 1) // 2)]:
                    bad.append((str(dt), got))
                self.ranks_observed = int(got[1])        # the sum of one `1` per rank, as RCCL delivered it
        if bad:
            raise _lib.DrsError("library-side RCCL all-reduce at world %d returned %s" % (W, bad))

    def _install_callback(self):
        self.collectives = "callback"
        ones = torch.ones(1, dtype=torch.int32, device=self.dev)
        self.comm.all_reduce_sum(ones)
        self.ranks_observed = int(ones.item())           # what the host's communicator sums one `1` per rank to
        spans = sorted((t.data_ptr(), t.data_ptr() + t.numel() * t.element_size(), t) for t in self._bufs.values())
        self._works = {}
        self._next_handle = 0

        def view(ptr, count):
            for lo, hi, t in spans:
                if lo <= ptr < hi:
                    off = (ptr - lo) // t.element_size()
                    return t[off:off + count]
            raise ValueError("all-reduce of memory outside the bound buffers")

        def allreduce(user, ptr, count, dtype, is_async, stream):
            try:
                v = view(ptr, count)
                if is_async:
                    hnd = self._next_handle
                    self._next_handle = (hnd + 1) % (1 << 30)
                    self._works[hnd] = self.comm.all_reduce_sum_async(v)
                    return hnd
                self.comm.all_reduce_sum(v)
                return 0
            except Exception as e:      # nothing may propagate into the library
                print("drs all-reduce callback failed:", repr(e))
                return -1

        def wait(user, hnd, stream):
            try:
                self.comm.wait([self._works.pop(hnd, None)])
                return 0
            except Exception as e:
                print("drs wait callback failed:", repr(e))
                return -1
        self._cb = (_lib.ALLREDUCE_FN(allreduce), _lib.WAIT_FN(wait))       # kept alive with the net
        _lib.call("drs_net_set_comm", self.h, self.comm.world, self.comm.rank, C.cast(self._cb[0], C.c_void_p), C.cast(self._cb[1], C.c_void_p), None)

    def set_two_streams(self, mode):
        """the backward pass of a step on two streams: None / -1 = by the library's rule (small steps), 0 never, 1 always
        (drs_net_set_two_streams; bitwise the same step in every mode)"""
        _lib.call("drs_net_set_two_streams", self.h, -1 if mode is None else int(mode))

    # ------------------------------------------------------------------ state the library owns
    @property
    def global_step(self):
        return int(_lib.query("drs_net_global_step", self.h, -1)) if self.h is not None else self._gs_pending

    @global_step.setter
    def global_step(self, v):
        if self.h is None:
            self._gs_pending = int(v)
        else:
            _lib.query("drs_net_global_step", self.h, int(v))

    @property
    def timer(self):
        return self.__dict__.get("_timer")

    @timer.setter
    def timer(self, v):
        if self.h is not None:
            _lib.call("drs_net_timing", self.h, 0 if v is None else 1)
        self.__dict__["_timer"] = None if v is None else EngineTimer(self)

    def learning_rate(self, lr0):
        return float(_lib.query("drs_net_learning_rate", self.h, float(lr0)))

    # ------------------------------------------------------------------ the three sess.run shapes
    def forward(self, B, S, want_logits=True, labels=False, acc_mask=False, ignore_label=-1):
        self._check(B, S)
        flags = (_lib.WANT_LOGITS if want_logits else 0) | (_lib.WITH_LABELS if labels else 0) | (_lib.USE_ACC_MASK if acc_mask else 0)
        _lib.call("drs_forward", self.h, B, S, flags, int(ignore_label), self._stream())
        M, K = B * S * S, self.plan.K
        return self.pred[:M].view(B, S, S), (self.logits[:M * K].view(B, S, S, K) if want_logits else None)

    def train_step(self, B, S, lr0, use_loss_mask=False, use_acc_mask=True, global_pixels=None, apply_update=True, want_logits=False):
        self._check(B, S)
        flags = ((_lib.USE_LOSS_MASK if use_loss_mask else 0) | (_lib.USE_ACC_MASK if use_acc_mask else 0) | (0 if apply_update else _lib.NO_UPDATE)
                 | (_lib.WANT_LOGITS if want_logits else 0))
        _lib.call("drs_train_step", self.h, B, S, float(lr0), flags, float(global_pixels) if global_pixels is not None else 0.0, self._stream())
        M, K = B * S * S, self.plan.K
        return dict(loss_parts=self.scalars[:2], pred=self.pred[:M].view(B, S, S), conf=self.conf.view(K, K))

    def apply_update(self, lr0):
        _lib.call("drs_apply_update", self.h, float(lr0), self._stream())
