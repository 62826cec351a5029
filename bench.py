#!/usr/bin/env python3
"""Headline benchmark: training patches/sec of dilated_grsl_rate8 (Dilated8Pooling) on 64x64x5 patches.

    python bench.py --gpus N --steps K --warmup W                      (N > 1: starts torch.distributed.run itself, as a child)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one global batch of 128 synthetic patches: device crop + augment +
normalise, forward, loss, backward, (sync-BN / gradient all-reduce when N > 1), momentum update, confusion
matrix.  The global batch is fixed, so N ranks take 128/N patches each ("scaling": "strong").  Inputs (the
2048x2048x5 tile, labels, instance table) are resident in HBM before the timed region.  Rank 0 prints ONE JSON
line with the contract's keys plus "roofline" (dominant kernel: the fp32-MFMA implicit-GEMM convolution, timed
live with HIP events on the launch stream), "kernels" (the same figures for every kernel family) and
"cpu_baseline" (the oracle's PyTorch-CPU port of the reference step, bounded sample, rank 0 at N=1 only), and at N=1
`extra.configs`: BASELINE.json's configs 2-5 on this one GPU (outside the timed region, ~15 s).  `--opt-in` adds
"opt_in_arithmetic": the same step on the split-bf16 convolution kernels (op-level path), never the headline `value`.

N > 1 cannot be lost to a hang in the collectives.  Every rank process of the launcher is a SUPERVISOR that never touches the
GPU: it starts the measuring process as a fresh child, relays its output, and watches its progress markers on stderr.  If a child
is silent for DRS_BENCH_WATCHDOG_S seconds (default 300; DRS_BENCH_WATCHDOG_STAGE_S = 120 once its imports are done) or dies, the supervisor kills it and starts a fresh child with the next
entry of DRS_BENCH_FALLBACKS (default "torch,async": the torch.distributed callback for every sum, then library-side RCCL in its
asynchronous two-communicator form; the first attempt is library-side RCCL inline, one communicator on the compute stream) on a
rendezvous port of its own; the JSON line then says which path ran and why
(`config.collectives`, `config.fallback_reason`).  A process that has touched the GPU is never re-executed.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver; must be set before the runtime starts

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

np = torch = None            # numpy / torch are imported by the measuring process only (load_numerics): a supervisor stays light


def load_numerics():
    global np, torch
    import numpy
    import torch as _torch
    np, torch = numpy, _torch


# ------------------------------------------------------------------------------------------------ launch plumbing (no GPU, no torch)
MARK = "DRS_BENCH_MARK"
FALLBACK_ENV = {"torch": {"DRS_COMM": "torch"}, "async": {"DRS_COMM": "rccl", "DRS_RCCL_ASYNC": "1"},
                "buckets": {"DRS_COMM": "rccl", "DRS_RCCL_BUCKETS": "2", "DRS_RCCL_ASYNC": "0"}}


def mark(stage, rank=None):
    """progress marker of a measuring process (stderr): what the supervisor's watchdog listens for"""
    sys.stderr.write("%s rank=%s stage=%s t=%.1f\n" % (MARK, os.environ.get("RANK", "0") if rank is None else rank, stage, time.time()))
    sys.stderr.flush()


_CHILDREN = []           # process groups this (GPU-free) parent started and has not reaped yet
_PENDING = {"lines": None}      # the finished default run's output while the optional second pass is still running (supervise)


def _flush_pending():
    """write the finished first run's line(s) exactly once, from wherever the supervisor leaves (normal return, signal handler)"""
    lines, _PENDING["lines"] = _PENDING["lines"], None
    if not lines:
        return
    for ln in lines:
        (sys.stdout if ln.lstrip().startswith("{") else sys.stderr).write(ln + "\n")
    try:
        sys.stdout.flush()
    except Exception:
        pass


def _kill_group(proc):
    import signal
    for sig, wait in ((signal.SIGTERM, 10.0), (signal.SIGKILL, 10.0)):
        try:
            os.killpg(proc.pid, sig)
        except (ProcessLookupError, PermissionError):
            pass
        try:
            proc.wait(timeout=wait)
            return
        except Exception:
            continue


def _die_with_parent():
    """preexec of every child: SIGTERM when the parent goes away however it goes (a SIGKILLed parent runs no handler).  The child is a
    session leader; a launcher child passes the signal on to its ranks, and those are supervisors with the handlers below."""
    import ctypes
    import signal
    try:
        ctypes.CDLL(None, use_errno=True).prctl(1, int(signal.SIGTERM), 0, 0, 0)       # PR_SET_PDEATHSIG
    except Exception:
        pass


def _install_cleanup_handlers():
    """SIGTERM / SIGINT / SIGHUP to a supervisor or to the plain parent: take the children's process groups down first (they run in
    sessions of their own, so nothing else would: `timeout ... python bench.py --gpus 8`, Ctrl-C, or the launcher tearing the other
    ranks down after one gave up would leave measuring processes on the GPUs -- a child hung in RCCL writes nothing and never sees
    EPIPE), then exit non-zero."""
    import signal

    def handler(signum, frame):
        _flush_pending()          # a completed measurement is never lost to a teardown during the optional second pass
        for proc in list(_CHILDREN):
            _kill_group(proc)
        # the plain parent relays its launcher's stdout at the end: a line a rank's supervisor got out while it was being torn down
        # (its own _flush_pending) is passed on here
        st = _PENDING.get("relay")
        if st is not None:
            time.sleep(0.3)       # (the pump thread reads what the dying children wrote)
            js = [ln for ln in "".join(st["out"]).splitlines() if ln.strip().startswith("{")]
            if js:
                sys.stdout.write(js[-1] + "\n")
                sys.stdout.flush()
        os._exit(128 + signum)
    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        try:
            signal.signal(sig, handler)
        except (ValueError, OSError):
            pass


def _run_watched(cmd, env, limit_s, need_marks=True, rendezvous_slack=1.0, wall_limit_s=None):
    """start `cmd` as a fresh process group, relay its stderr (watching the markers) and collect its stdout.
    Returns (rc or None if killed by the watchdog, stdout text, last marker stage, the silence limit that applied last).
    `rendezvous_slack` multiplies the limit until the process group is up: the ranks of a fall-back attempt arrive up to one watchdog
    period apart (each supervisor times out alone).  `wall_limit_s`: a cap on the child's TOTAL run time, markers or not.  However
    this function is left (return, exception, signal handler), the child's process group does not outlive it."""
    import subprocess
    import threading
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True, preexec_fn=_die_with_parent)
    _CHILDREN.append(proc)
    state = dict(last=time.time(), stage="started", out=[])
    t_start = time.time()
    if not need_marks:
        _PENDING["relay"] = state      # (plain_parent's one child: see the signal handler)

    def pump_err():
        for raw in iter(proc.stderr.readline, b""):
            line = raw.decode("utf-8", "replace")
            if MARK in line:
                state["last"] = time.time()
                try:
                    state["stage"] = line.split("stage=", 1)[1].split(" t=")[0].strip()
                except IndexError:
                    pass
            sys.stderr.write(line)
            sys.stderr.flush()

    def pump_out():
        for raw in iter(proc.stdout.readline, b""):
            state["out"].append(raw.decode("utf-8", "replace"))
    th = [threading.Thread(target=pump_err, daemon=True), threading.Thread(target=pump_out, daemon=True)]
    for t in th:
        t.start()
    killed = False
    lim = limit_s
    try:
        while True:
            try:
                proc.wait(timeout=1.0)
                break
            except subprocess.TimeoutExpired:
                # until its imports are done a process may be paging the image in (minutes on a fresh box): the long limit; after that
                # every stage is seconds long: the short one (rendezvous of a fall-back attempt: the ranks arrive up to a period apart)
                stage_s = float(os.environ.get("DRS_BENCH_WATCHDOG_STAGE_S", "120"))
                lim = limit_s if state["stage"] == "started" else min(limit_s, stage_s)
                if state["stage"] in ("started", "imports done"):
                    lim *= rendezvous_slack
                if need_marks and time.time() - state["last"] > lim:
                    sys.stderr.write("bench.py watchdog: no progress marker for %.0f s after stage '%s': killing the process group\n" % (lim, state["stage"]))
                    killed = True
                    break
                if wall_limit_s is not None and time.time() - t_start > wall_limit_s:
                    sys.stderr.write("bench.py watchdog: %.0f s of wall time used up at stage '%s': killing the process group\n" % (wall_limit_s, state["stage"]))
                    lim = wall_limit_s
                    killed = True
                    break
    finally:
        # the leader may be gone while members of its group live on (a launcher's ranks): signal the GROUP whatever the leader's state
        if killed or proc.poll() is None:
            _kill_group(proc)
        else:
            import signal
            try:
                os.killpg(proc.pid, signal.SIGKILL)
            except (ProcessLookupError, PermissionError):
                pass
        if proc in _CHILDREN:
            _CHILDREN.remove(proc)
    for t in th:
        t.join(timeout=5.0)
    return (None if killed else proc.returncode), "".join(state["out"]), state["stage"], lim


def _fallback_port(port0, i):
    """rendezvous port of fall-back attempt i: every supervisor of the launch must arrive at the same one without talking to the others.
    Rank 0's supervisor checks the derived port (port0 + 101 i), takes a free one if that is taken, and publishes its choice in a file
    keyed by the launcher (the supervisors' common parent) that the other ranks wait for; without the file (another host's /tmp, a
    read-only one) everybody falls back to the derived port."""
    import socket
    import tempfile
    derived = port0 + 101 * i
    # keyed by the launcher's pid AND a per-launch token (a reused pid with the same port must not find an earlier launch's file)
    path = os.path.join(tempfile.gettempdir(), "drs_bench_%d_%s_%d_%d.port" % (os.getppid(), _launch_token(), port0, i))
    if os.environ.get("RANK", "0") == "0":
        try:
            os.unlink(path)          # whatever is there is not this attempt's
        except OSError:
            pass
        port = derived
        try:
            with socket.socket() as so:
                so.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                so.bind(("127.0.0.1", derived))
        except OSError:
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                port = so.getsockname()[1]
        try:
            with open(path + ".tmp", "w") as f:
                f.write(str(port))
            os.replace(path + ".tmp", path)
        except OSError:
            return derived
        return port
    # rank 0's supervisor may arrive a whole watchdog period (x its rendezvous slack) after this one: wait that long before giving up
    stage_s = float(os.environ.get("DRS_BENCH_WATCHDOG_STAGE_S", "120"))
    t_end = time.time() + float(os.environ.get("DRS_BENCH_PORT_WAIT_S", str(2.5 * stage_s)))
    while time.time() < t_end:
        try:
            return int(open(path).read().strip())
        except (OSError, ValueError):
            time.sleep(0.2)
    return derived


def supervise(argv):
    """one rank of the launcher: start the measuring process as a child, fall back on a hang or a crash.  Never touches the GPU."""
    limit = float(os.environ.get("DRS_BENCH_WATCHDOG_S", "300"))
    chain = [("default", {})] + [(k, FALLBACK_ENV[k]) for k in os.environ.get("DRS_BENCH_FALLBACKS", "torch,async").split(",") if k in FALLBACK_ENV]
    port0 = int(os.environ.get("MASTER_PORT", "29500"))
    reason = ""
    _install_cleanup_handlers()
    for i, (label, extra) in enumerate(chain):
        env = dict(os.environ)
        env.update(extra)
        env.update(DRS_BENCH_CHILD="1", DRS_BENCH_ATTEMPT=label, DRS_BENCH_FALLBACK_REASON=reason)
        # first contact with RCCL at N > 1 must be cheap to diagnose: warnings of every attempt go to a file per attempt and rank
        # (RCCL itself expands %h / %p); after a failed attempt its tail travels in the next attempt's line and in the failure line
        if env.get("NCCL_DEBUG", "VERSION").upper() == "VERSION":      # (this image exports VERSION: the banner only)
            env["NCCL_DEBUG"] = "WARN"
        if "NCCL_DEBUG_FILE" not in os.environ:
            env["NCCL_DEBUG_FILE"] = os.path.join(_tmpdir(), "drs_bench_nccl_%d_%s_rank%s.%%h.%%p.log" % (os.getppid(), label, os.environ.get("RANK", "0")))
        if i > 0:
            # a rendezvous of its own: the launcher's store still holds the keys of the attempt that was killed.  Rank 0's child hosts it.
            env.update(MASTER_PORT=str(_fallback_port(port0, i)), TORCHELASTIC_USE_AGENT_STORE="False")
        rc, out, stage, lim = _run_watched([sys.executable, os.path.abspath(__file__)] + argv, env, limit, rendezvous_slack=1.0 if i == 0 else 2.5)
        if rc == 0:
            # stdout carries the ONE JSON line (rank 0's) and nothing else: whatever a library printed to the child's descriptor 1 goes to stderr
            lines = out.splitlines()
            _PENDING["lines"] = lines       # from here on the line gets out whatever happens (signal handler, exception below)
            if label == "default" and os.environ.get("DRS_BENCH_SECOND_PASS", "1") != "0":
                try:
                    _PENDING["lines"] = _second_pass(argv, port0, lines)
                except Exception as e:      # the optional pass must never cost the first one
                    sys.stderr.write("bench.py supervisor: second pass failed in the supervisor: %r\n" % (e,))
            _flush_pending()
            _cleanup_port_files(port0)
            return 0
        reason = ("timeout: silent for %.0f s after stage '%s'" % (lim, stage)) if rc is None else ("exit code %s after stage '%s'" % (rc, stage))
        reason = "%s attempt: %s" % (label, reason)
        tail = _nccl_debug_tail(env.get("NCCL_DEBUG_FILE"))
        if tail:
            os.environ["DRS_BENCH_NCCL_TAIL"] = tail      # the next attempt's line carries it (extra.first_contact.failed_attempt_nccl_tail)
        sys.stderr.write("bench.py supervisor (rank %s): %s%s\n" % (os.environ.get("RANK", "?"), reason, "; falling back" if i + 1 < len(chain) else "; giving up"))
    _cleanup_port_files(port0)
    if os.environ.get("RANK", "0") == "0":
        sys.stdout.write(failure_line("every attempt failed; last: " + reason, nccl_tail=os.environ.get("DRS_BENCH_NCCL_TAIL")))
        sys.stdout.flush()
    else:
        # the launcher tears every rank down as soon as ONE exits non-zero: let rank 0's line get out first
        time.sleep(float(os.environ.get("DRS_BENCH_FAIL_LINGER_S", "5")))
    return 1


def _second_pass(argv, port0, first_lines):
    """After a default run that WORKED (library-side RCCL, inline form) every supervisor starts one more fresh measuring process with
    DRS_RCCL_BUCKETS=2 -- the inline form with the gradient buffer as two overlapped all-reduces on a side stream -- so that the one
    multi-GPU lease a round gets returns both numbers.  Its line goes INTO the first line (`extra.second_pass`); whatever happens to
    it (a hang is cut after DRS_BENCH_SECOND_PASS_LIMIT_S of silence, default 120) the first line stands as it is.  The headline
    `value` is never replaced."""
    env = dict(os.environ)
    env.update(FALLBACK_ENV["buckets"])
    env.update(DRS_BENCH_CHILD="1", DRS_BENCH_ATTEMPT="buckets", DRS_BENCH_FALLBACK_REASON="", DRS_BENCH_IS_SECOND_PASS="1",
               MASTER_PORT=str(_fallback_port(port0, 7)), TORCHELASTIC_USE_AGENT_STORE="False")
    limit = float(os.environ.get("DRS_BENCH_SECOND_PASS_LIMIT_S", "120"))
    wall = float(os.environ.get("DRS_BENCH_SECOND_PASS_WALL_S", "240"))       # silence AND total time are bounded: the driver's own limit is not ours to spend
    rc, out, stage, lim = _run_watched([sys.executable, os.path.abspath(__file__)] + argv + ["--no-cpu-baseline", "--no-opt-in", "--no-size-table", "--no-configs"],
                                       env, limit, rendezvous_slack=1.5, wall_limit_s=wall)
    if os.environ.get("RANK", "0") != "0":
        return first_lines
    res = None
    if rc == 0:
        for ln in out.splitlines():
            if ln.lstrip().startswith("{"):
                try:
                    d = json.loads(ln)
                    res = {k: d.get(k) for k in ("value", "unit", "ms_per_step", "median_step_ms", "steps", "warmup", "config", "kernels", "final_loss")}
                    res["per_rank_ms"] = (d.get("extra") or {}).get("per_rank_ms")
                except ValueError:
                    pass
    if res is None:
        res = {"error": ("timeout: silent for %.0f s after stage '%s'" % (lim, stage)) if rc is None else ("exit code %s after stage '%s'" % (rc, stage))}
    res["note"] = "second timed pass of the same command with DRS_RCCL_BUCKETS=2 (a fresh process per rank); never the headline value"
    merged = []
    for ln in first_lines:
        if ln.lstrip().startswith("{"):
            try:
                d = json.loads(ln)
                d.setdefault("extra", {})
                if d["extra"] is None:
                    d["extra"] = {}
                d["extra"]["second_pass"] = res
                ln = json.dumps(d)
            except ValueError:
                pass
        merged.append(ln)
    return merged


def failure_line(error, nccl_tail=None):
    return json.dumps({"metric": "training patches/sec dilated_grsl_rate8 64x64x5", "value": None, "unit": "patches/s",
                       "n_gpus": int(os.environ.get("WORLD_SIZE", "1")), "error": error,
                       "extra": {"first_contact": {"failed_attempt_nccl_tail": nccl_tail}}}) + "\n"


def _tmpdir():
    import tempfile
    return tempfile.gettempdir()


def _launch_token():
    """what tells one launch's files from another's with a reused pid: the launcher's run id where there is one"""
    import hashlib
    start = ""
    try:       # the launcher (the supervisors' common parent): its start time in clock ticks since boot -- the same for every rank, new per launch
        start = open("/proc/%d/stat" % os.getppid()).read().rsplit(")", 1)[1].split()[19]
    except (OSError, IndexError):
        pass
    return hashlib.sha1((os.environ.get("TORCHELASTIC_RUN_ID", "") + "|" + os.environ.get("MASTER_PORT", "") + "|" + start).encode()).hexdigest()[:10]


def _nccl_debug_tail(pattern, nlines=12):
    """the last lines RCCL wrote (NCCL_DEBUG=WARN) into this rank's debug file(s) of a failed attempt, as one short string"""
    import glob
    if not pattern:
        return None
    out = []
    for f in sorted(glob.glob(pattern.replace("%h", "*").replace("%p", "*"))):
        try:
            out.extend(ln.rstrip() for ln in open(f, errors="replace").read().splitlines()[-nlines:])
        except OSError:
            continue
    text = " | ".join(ln for ln in out[-nlines:] if ln)
    return text[-1500:] if text else None


def _cleanup_port_files(port0):
    import glob
    for f in glob.glob(os.path.join(_tmpdir(), "drs_bench_%d_%s_%d_*.port*" % (os.getppid(), _launch_token(), port0))):
        try:
            os.unlink(f)
        except OSError:
            pass


def plain_parent(args, argv):
    """`python bench.py --gpus N` without a launcher: start torch.distributed.run as a fresh child (whose ranks supervise themselves),
    relay its one JSON line and its exit code.  Never touches the GPU."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    _install_cleanup_handlers()
    rc, out, _, _ = _run_watched(cmd, dict(os.environ), 0.0, need_marks=False)
    lines = [ln for ln in out.splitlines() if ln.strip().startswith("{")]
    if lines:
        sys.stdout.write(lines[-1] + "\n")
    elif rc != 0:
        # (the launcher can take rank 0 down before its failure line is out: there is still exactly one line, and it says what happened)
        os.environ["WORLD_SIZE"] = str(args.gpus)
        sys.stdout.write(failure_line("every attempt failed; the launcher exited with code %s before rank 0 reported" % rc))
    sys.stdout.flush()
    return rc if rc is not None else 1


def selftest_worker(args):
    """DRS_BENCH_SELFTEST=1: the launch plumbing alone, on CPU (tests/test_bench_launch.py): gloo process group, one barrier, the
    markers, one JSON line from rank 0.  DRS_BENCH_FAKE_HANG=<attempt label> makes that attempt hang before its warm-up marker."""
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    attempt = os.environ.get("DRS_BENCH_ATTEMPT", "default")
    sys.stdout.flush()
    json_fd = os.dup(1)          # gloo prints its connection lines to descriptor 1
    os.dup2(2, 1)
    mark("imports done")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
        dist.barrier()
    mark("process group up")
    if attempt in os.environ.get("DRS_BENCH_FAKE_HANG", "").split(","):
        time.sleep(3600)
    if attempt in os.environ.get("DRS_BENCH_FAKE_CRASH", "").split(","):
        sys.exit(7)
    mark("warm-up done")
    ones = None
    if world > 1:
        import torch as _t
        ones = _t.ones(1, dtype=_t.int32)
        dist.all_reduce(ones)
        dist.barrier()
    if rank == 0:
        os.write(json_fd, (json.dumps({"metric": "selftest", "value": 1.0, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "config": {"collectives": collectives_label("selftest"), "fallback_reason": os.environ.get("DRS_BENCH_FALLBACK_REASON") or None,
                                     "ranks_observed": int(ones.item()) if ones is not None else 1,
                                     "env": {k: os.environ.get(k) for k in ("DRS_COMM", "DRS_RCCL_ASYNC", "DRS_RCCL_BUCKETS", "MASTER_PORT")}}}) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()
    return 0


def collectives_label(base):
    """`config.collectives` of the JSON line: the path the step's sums took, and whether it is a fall-back of the supervisor"""
    attempt, reason = os.environ.get("DRS_BENCH_ATTEMPT", "default"), os.environ.get("DRS_BENCH_FALLBACK_REASON", "")
    if attempt != "default" and reason:
        return "%s (fallback after %s)" % (base, "timeout" if "timeout" in reason else "failure")
    return base


NET, CHANNELS, CLASSES = "dilated_grsl_rate8", 5, 6
GLOBAL_BATCH, PATCH = 128, 64
TILE = 2048
LR, WD = 0.01, 0.005
PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 = fp32 vector rate
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec peak
# arithmetic of the convolution kernels (net.ARITH_TERMS): MFMA peak it is priced against, MFMA products issued per
# algorithmic multiply, the dtype string of the JSON line and the dominant kernel's name
ARITH = {
    "f32": dict(peak=PEAK_FP32_MFMA_TFLOPS, products=1, dtype="f32", kernel="conv_dma_kernel (fwd + dgrad launches; conv1: conv_igemm_kernel)",
                pmc_key="conv fwd+dgrad (all)"),
    "bf16x3": dict(peak=PEAK_BF16_MFMA_TFLOPS, products=3, kernel="conv_split_dma_kernel (fwd + dgrad launches)",
                   dtype="bf16x3 (fp32 operands as 2 bf16 terms, 3 bf16 MFMA products per multiply, fp32 accumulate)",
                   pmc_key="conv_split_dma_kernel (all)"),
    "bf16x6": dict(peak=PEAK_BF16_MFMA_TFLOPS, products=6, kernel="conv_split_dma_kernel (fwd + dgrad launches)",
                   dtype="bf16x6 (fp32 operands as 3 bf16 terms, 6 bf16 MFMA products per multiply, fp32 accumulate)",
                   pmc_key="conv_split_dma_kernel (all)"),
}


def pmc_traffic(arith):
    """HBM-side bytes per launch of the dominant conv kernel from the committed rocprofv3 --pmc passes of this same command
    (profiles/rNN/pmc_traffic[_<arith>].json, written by tools/pmc_summary.py); None when no profile is committed."""
    import glob
    name = "pmc_traffic.json" if arith == "f32" else "pmc_traffic_%s.json" % arith
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", name)))
    if not files:
        return None
    for f in reversed(files):           # newest round that has the key (older rounds named the kernel family differently)
        try:
            return json.load(open(f))[ARITH[arith]["pmc_key"]]["traffic_bytes_per_launch"]
        except (KeyError, ValueError):
            continue
    return None


def allowed_cores():
    """CPU cores this process may really use: its affinity mask, cut to the cgroup's CPU-time quota where one is set (a GPU box
    shows every core of the host in the mask but grants one GPU's share of CPU time: more threads than that only thrash)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    note = "affinity mask: %d" % n
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1") and float(quota) > 0:
                q = max(1, int(float(quota) / period + 0.5))
                if q < n:
                    note += ", cgroup CPU quota: %d" % q
                    n = q
            break
        except (OSError, ValueError, IndexError):
            continue
    return n, note


def cpu_baseline(batch=GLOBAL_BATCH, warmup=2, timed=5):
    """The reference step restated with PyTorch-CPU fp32 ops (oracle/torch_ref.py) + the reference's host pipeline (numpy crop /
    rotate / noise / flip, normalise of bands 0..2, confusion matrix), on this host's cores, as SURVEY.md 8(d) specifies it: the
    bench's own workload (same net, 64x64x5 patches of the same 2048 x 2048 synthetic tile, same instances), `warmup` + `timed`
    steps at the GPU line's own batch of 128 (2 warm-up + 5 timed; a CPU step is ~10 s long, so ~70 s in all, each step
    leaving a progress marker), every core
    the process is allowed (allowed_cores: affinity mask cut to the cgroup's CPU quota), median step; plus a 1-thread figure from one step
    of 8 patches.  Both are stated in `sample`."""
    from oracle.torch_ref import TorchNet
    from oracle import host_ref as H
    from oracle.tf_ops import OracleNet
    from drs_amd.synthetic import make_tile, grid_instances
    ncores, core_note = allowed_cores()
    torch.set_num_threads(ncores)
    tile, lab = make_tile(TILE, TILE, CHANNELS, CLASSES, seed=1234)
    inst = grid_instances(TILE, TILE, PATCH, 25, GLOBAL_BATCH * 100, seed=0)
    mean = tile[:, :, :3].mean(axis=(0, 1)).tolist() + [0, 0]
    std = tile[:, :, :3].std(axis=(0, 1)).tolist() + [1, 1]
    o = OracleNet(NET, CHANNELS, CLASSES, dtype=np.float32, seed=42)
    net = TorchNet(NET, CHANNELS, CLASSES, params=o.p, dtype=torch.float32)
    track = np.zeros((CLASSES, CLASSES), dtype=np.uint32)
    np.random.seed(0)

    def step(i, B):
        rows = inst[(i * B) % (len(inst) - B):(i * B) % (len(inst) - B) + B]
        t0 = time.time()
        x, y, m = H.dynamically_create_patches([tile], [lab], rows, PATCH, is_train=True)
        H.normalize_images(x, mean, std)
        _, pred = net.train_step(x.astype(np.float32), y, LR, WD)
        H.calc_accuracy_by_crop(y, pred, track, m, CLASSES)
        return time.time() - t0
    for i in range(warmup):
        step(i, batch)
        mark("cpu baseline warm-up %d" % i)
    times = []
    for i in range(timed):
        times.append(step(warmup + i, batch))
        mark("cpu baseline step %d" % i)
    med = float(np.median(times))
    torch.set_num_threads(1)
    b1 = 8
    one = step(0, b1)
    torch.set_num_threads(ncores)
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return dict(value=round(batch / med, 3), unit="patches/s", cores=ncores, cores_note=core_note, kind="port", cpu_model=model,
                value_1_thread=round(b1 / one, 3),
                sample="%d timed steps after %d warm-up, batch %d (= the GPU line's), dilated_grsl_rate8 / 64x64x5 patches of the 2048x2048 "
                       "tile, host crop+augment+normalise+confusion included; median step %.2f s (min %.2f, max %.2f); 1-thread figure: one "
                       "step of %d patches, %.2f s" % (timed, warmup, batch, med, min(times), max(times), b1, one))


def _lib_query(name, *args):
    from drs_amd import _lib
    return _lib.query(name, *args)


def executed_fraction(plan, B, S):
    """Share of the algorithmic multiply-adds of the forward / input-gradient launches that the kernels really issue: filter-tap
    rows that meet only the zero halo for a whole M tile are skipped (drs_common.hpp live_tap_rows) in the launches the library says
    (drs_conv_halo_skip: plain launches of >= 4096 tiles, or any plain launch that takes the full-tiles-first order).  Which rows a
    tile skips is mirrored here: M tiles of 128 pixels, whole tap rows, a tile that crosses an image boundary keeps every row."""
    M = B * S * S
    tot = live = 0.0
    for i, L in enumerate(plan.layers):
        # forward (N = Cout, contraction over Cin); input gradient (none for conv1; N = Cin, contraction over Cout)
        for pad, cout, gemm_k, on in ((L.pad_b, L.cout, L.cin_k, True), (L.pad_a, L.cin, L.cout, i > 0)):
            if not on:
                continue
            work = float(L.k * L.k * L.cin * L.cout)
            skip = bool(_lib_query("drs_conv_halo_skip", B, S, L.k, L.rate, pad, gemm_k, cout))      # the library's own rule (conv_mfma.hip)
            frac = 1.0
            if skip:
                rows = 0
                ntile = (M + 127) // 128
                for t in range(ntile):
                    p0, p1 = t * 128, min(t * 128 + 127, M - 1)
                    b0, b1 = p0 // (S * S), p1 // (S * S)
                    y0, y1 = (p0 % (S * S)) // S, (p1 % (S * S)) // S
                    if b0 != b1:
                        y0, y1 = 0, S - 1
                    a = pad - y1
                    lo = (a + L.rate - 1) // L.rate if a > 0 else 0
                    hi = min((S - 1 - y0 + pad) // L.rate + 1, L.k)
                    if lo >= hi:
                        lo, hi = 0, L.k
                    rows += hi - lo
                frac = rows / float(ntile * L.k)
            tot += work
            live += work * frac
    return live / tot


def per_rank_size_table(dev, pool, mean, std, sizes=(25, 35, 45, 55, 64, 65, 75, 85), B=16, steps=8):
    """what one rank of the 8-GPU run of BASELINE configs[2] sees (batch 128 / 8 = 16 patches, `uniform` over [25, 85],
    isprs:1727-1737): ms per step and the convolution families' share of the fp32 MFMA roof, per patch side (N = 1 only, outside
    the timed region).  `weighted_patches_per_s` = sizes drawn uniformly: patches / summed step time."""
    from drs_amd.net import DilatedNet, KernelTimer
    from drs_amd import patches as P
    from drs_amd.synthetic import grid_instances
    net = DilatedNet(NET, CHANNELS, CLASSES, WD, b_max=B, s_max=max(sizes), device=dev, seed=42)
    rows_out, tot, headline = [], 0.0, None
    for S in sizes:
        inst = grid_instances(TILE, TILE, S, 25, 2048, seed=S)

        def step(i):
            r = inst[(i * B) % 2000:(i * B) % 2000 + B]
            aug = P.draw_augmentation(r, S, CHANNELS, noise="device")
            P.crop_to_net(net, pool, r, S, mean, std, aug)
            return net.train_step(B, S, LR)
        for i in range(3):
            step(i)
        dts = []
        for _ in range(3):               # three blocks of `steps`, the median block: one host hiccup in a 20..100 ms block is not the GPU's step
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                step(i)
            torch.cuda.synchronize()
            dts.append((time.perf_counter() - t0) / steps)
        dt = sorted(dts)[1]
        net.timer = KernelTimer()
        for i in range(3):
            step(i)
        summ = net.timer.summary()
        net.timer = None
        fd = [summ[k] for k in ("conv_fwd", "conv_dgrad") if k in summ]
        fd_tf = sum(d["work"] for d in fd) / (sum(d["ms"] for d in fd) * 1e-3) / 1e12
        wg_tf = summ["conv_wgrad"]["work"] / (summ["conv_wgrad"]["ms"] * 1e-3) / 1e12
        row = dict(S=S, ms_per_step=round(dt * 1e3, 3), patches_per_s=round(B / dt, 1),
                   fwd_dgrad_frac=round(fd_tf / PEAK_FP32_MFMA_TFLOPS, 3), wgrad_frac=round(wg_tf / PEAK_FP32_MFMA_TFLOPS, 3))
        if S == PATCH:
            # one rank of the 8-GPU run of the HEADLINE (128 / 8 patches of 64 x 64): every family's ms per step, so that the driver's
            # 8-GPU line (whose own `kernels` are rank 0's at this batch) can be held against what one GPU alone does at that batch
            headline = dict(row, kernels_ms={k: round(d["ms"] / 3, 4) for k, d in sorted(summ.items())},
                            ideal_speedup_before_wire=None)
            continue          # (not a size configs[2] draws from: not part of the weighted figure)
        rows_out.append(row)
        tot += dt
    del net
    torch.cuda.empty_cache()
    return dict(local_batch=B, rows=rows_out, weighted_patches_per_s=round(B * len(rows_out) / tot, 1), headline_shape=headline,
                note="one rank's step at the per-rank batch of an 8-GPU run, no collectives; stream-K convolutions below 4096 tiles")


def baseline_configs(dev, steps=20):
    """BASELINE.json configs[1..4] on this one GPU, outside the timed region (the same loops and draws as
    tools/bench_configs.py, so the driver's line and the builder's logs can be held against each other):
      config 2  dilated_grsl (Dilated6Pooling), single_fixed 64 x 64, 5 bands, batch 64, 2048^2 tile       isprs:962-993
      config 3  dilated_grsl_rate8, `uniform` over [25, 85] (any integer side per step, isprs:1727-1737), batch 128
      config 4  dilated_icpr_rate6_densely (DenseDilated6), `multinomial` over {25, 50, 75, 100}, 4 bands, 2 classes, batch 128
      config 5  dilated_grsl_rate8 sliding-window inference of a 6000 x 6000 x 5 mosaic, 64 x 64 windows at stride 32, overlap-add of
                the logits, arg-max map (isprs:1241-1284) + a checksum of that map
    Each: the rate, and its share of the fp32 MFMA ceiling (algorithmic flops of the steps really run / time / 157.3 TFLOP/s)."""
    from drs_amd.net import DilatedNet
    from drs_amd import loops, patches as P
    from drs_amd.synthetic import make_tile, grid_instances
    out = {}

    def train_rate(key, net_type, ch, K, B, tile_side, draw, label):
        tile, lab = make_tile(tile_side, tile_side, ch, K, seed=1234)
        pool = P.TilePool([tile], [lab], dev, dtype=np.float64)
        mean, std = tile[:, :, :3].mean(axis=(0, 1)).tolist(), tile[:, :, :3].std(axis=(0, 1)).tolist()
        np.random.seed(11)
        sizes = [draw() for _ in range(steps + 5)]
        s_max = max(sizes)
        net = DilatedNet(net_type, ch, K, WD, b_max=B, s_max=s_max, device=dev, seed=42)
        inst = grid_instances(tile_side, tile_side, s_max, 25, max(B * 64, 2 * B), seed=0)

        def step(i):
            sd = sizes[i]
            o = (i * B) % max(1, len(inst) - B)
            rows = inst[o:o + B]
            aug = P.draw_augmentation(rows, sd, ch, noise="device")
            P.crop_to_net(net, pool, rows, sd, mean, std, aug)
            return net.train_step(B, sd, LR)
        for i in range(5):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(5, steps + 5):
            res = step(i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        px = sum(B * sd * sd for sd in sizes[5:])
        flops = 3 * 2 * net.plan.mac_per_pixel() * px
        out[key] = dict(workload=label, value=round(B * steps / dt, 1), unit="patches/s", ms_per_step=round(1e3 * dt / steps, 3), steps=steps, warmup=5,
                        batch=B, mean_side=round(float(np.mean(sizes[5:])), 1), trained_mpx_per_s=round(px / dt / 1e6, 1),
                        tflops=round(flops / dt / 1e12, 1), fp32_ceiling_frac=round(flops / dt / (PEAK_FP32_MFMA_TFLOPS * 1e12), 4),
                        final_loss=round(net.loss_value(res["loss_parts"]), 5))
        del net, pool
        torch.cuda.empty_cache()
        mark("%s done" % key)
    train_rate("config2", "dilated_grsl", 5, 6, 64, 2048, lambda: 64,
               "dilated_grsl (Dilated6Pooling) training, single_fixed 64x64, 5 bands, batch 64, 2048x2048 tile")
    v3 = [25, 45, 65, 85]
    train_rate("config3", NET, CHANNELS, CLASSES, 128, 2048, lambda: P.draw_patch_size("uniform", v3)[0],
               "dilated_grsl_rate8 training, `uniform` over [25, 85] (any integer side per step), 5 bands, batch 128 on ONE GPU")
    v4 = [25, 50, 75, 100]
    probs = P.define_multinomial_probs(v4)
    train_rate("config4", "dilated_icpr_rate6_densely", 4, 2, 128, 500, lambda: P.draw_patch_size("multinomial", v4, probs)[0],
               "dilated_icpr_rate6_densely (DenseDilated6) training, `multinomial` over {25,50,75,100}, update_type=loss sizes, 4 bands, 2 classes, batch 128 on ONE GPU")
    # ---- config 5: the mosaic is made ON the device (SURVEY 8d's recipe: bands 0-2 U[0,1) under a 9 x 9 box filter, band 3 U[0,1), band 4
    # 0.2 U[0,1); a host-side make_tile of 36 M pixels takes ~40 s and adds nothing to a forward-only timing)
    n, S, Bw = 6000, 64, 256
    g0 = torch.Generator(device=dev).manual_seed(5)
    m = torch.rand(5, n, n, device=dev, generator=g0)
    m[:3] = torch.nn.functional.avg_pool2d(m[:3].unsqueeze(0), 9, stride=1, padding=4, count_include_pad=False)[0]
    m[4] *= 0.2
    mosaic = m.permute(1, 2, 0).contiguous().reshape(-1)
    mean = m[:3].mean(dim=(1, 2)).tolist()
    std = m[:3].std(dim=(1, 2)).tolist()
    del m
    pool = P.TilePool([np.zeros((S, S, CHANNELS), dtype=np.float32)], None, dev, dtype=np.float32)      # shell; the mosaic is on the device already
    pool.tiles, pool.labels = mosaic, torch.zeros(n * n, dtype=torch.uint8, device=dev)
    pool.h, pool.w = [n], [n]
    pool.tile_h = torch.tensor([n], dtype=torch.int32, device=dev)
    pool.tile_w = torch.tensor([n], dtype=torch.int32, device=dev)
    net = DilatedNet(NET, CHANNELS, CLASSES, WD, b_max=Bw, s_max=S, device=dev, seed=42)
    # a random-init net in eval mode (moving statistics 0 / 1) saturates on one class: take the moving statistics from the batch
    # statistics of 256 windows spread over the mosaic (one train-mode pass, no update), so that the stitched map -- and its checksum -- is
    # a map of six classes
    nh, nw = P.window_counts(n, n, S, S // 2)
    spread = np.linspace(0, nh * nw - 1, Bw).astype(np.int64)
    allpos = np.stack([np.minimum((spread // nw) * (S // 2), n - S), np.minimum((spread % nw) * (S // 2), n - S)], axis=1)
    P.crop_to_net(net, pool, np.concatenate([np.zeros((Bw, 1), dtype=np.int64), allpos], axis=1), S, mean, std)
    net.train_step(Bw, S, 0.0, apply_update=False)
    torch.cuda.synchronize()
    for i, L in enumerate(net.plan.layers):
        mr = net.mean_rstd[i].cpu().numpy().reshape(L.cout, 2).astype(np.float64)
        net.set_variable(L.name + "/moving_mean", mr[:, 0])
        net.set_variable(L.name + "/moving_variance", np.maximum(1.0 / mr[:, 1] ** 2 - 1e-3, 1e-6))
    P.crop_to_net(net, pool, np.concatenate([np.zeros((Bw, 1), dtype=np.int64), P.window_positions(n, n, S, S // 2, 0, Bw)], axis=1), S, mean, std)
    net.forward(Bw, S)                       # code objects loaded outside the timing
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pred, _ = loops.predict_tile(net, pool, 0, S, Bw, mean, std)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nwin = nh * nw
    flops = 2.0 * net.plan.mac_per_pixel() * nwin * S * S
    hist = torch.bincount(pred.reshape(-1).long(), minlength=CLASSES)
    idx = torch.arange(n * n, device=dev, dtype=torch.int64)
    checksum = int(((pred.reshape(-1).long() + 1) * (idx % 65521 + 1)).sum().item() % (1 << 61))
    out["config5"] = dict(workload="dilated_grsl_rate8 sliding-window inference, 6000x6000x5 synthetic mosaic, 64x64 windows at stride 32, "
                                   "overlap-add of logits + arg-max map, ONE GPU (window batches of 256)",
                          value=round(dt, 3), unit="s", windows=nwin, window_mpx_per_s=round(nwin * S * S / dt / 1e6, 1),
                          map_mpx_per_s=round(n * n / dt / 1e6, 2), fp32_floor_s=round(flops / (PEAK_FP32_MFMA_TFLOPS * 1e12), 3),
                          fp32_ceiling_frac=round(flops / dt / (PEAK_FP32_MFMA_TFLOPS * 1e12), 4),
                          map_class_histogram=[int(v) for v in hist.tolist()],
                          map_checksum="sum((label+1) * (flat_index %% 65521 + 1)) mod 2^61 = %d" % checksum)
    del net, pool, pred, mosaic
    torch.cuda.empty_cache()
    mark("config5 done")
    out["note"] = ("BASELINE.json configs[1..4] on one GPU, each outside the headline's timed region; same loops, seeds and draws as "
                   "tools/bench_configs.py (config 5's mosaic is synthesised on the device here)")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)     # 100 x 53 ms: long enough for a driver that samples GPU activity every few seconds
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--opt-in", action="store_true", help="also time the step on the opt-in split-bf16 convolution kernels (op-level path; never the "
                                                         "headline and not credited: narrower than the reference's fp32 -- off by default since round 6)")
    ap.add_argument("--no-opt-in", action="store_true", help="(accepted for older command lines; the opt-in pass is off unless --opt-in)")
    ap.add_argument("--no-size-table", action="store_true", help="skip the per-rank patch-size table (N = 1 only)")
    ap.add_argument("--no-configs", action="store_true", help="skip BASELINE configs 2-5 beside the headline (N = 1 only, ~15 s)")
    ap.add_argument("--arith", choices=sorted(ARITH), default="f32",
                    help="arithmetic of the convolution kernels: exact fp32 MFMA (default) or split-bf16 (conv_split.hip)")
    args = ap.parse_args()
    ar = ARITH[args.arith]
    # ---- which process is this?  (decided before anything imports torch or touches the GPU)
    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    child = os.environ.get("DRS_BENCH_CHILD") == "1"
    if not launched and args.gpus > 1:
        sys.exit(plain_parent(args, sys.argv[1:]))                # `python bench.py --gpus N`: start the launcher as a child
    if launched and not child and int(os.environ["WORLD_SIZE"]) > 1 and os.environ.get("DRS_BENCH_SUPERVISE", "1") != "0":
        sys.exit(supervise(sys.argv[1:]))                         # a rank of the launcher: supervise a fresh measuring process
    if os.environ.get("DRS_BENCH_SELFTEST") == "1":
        sys.exit(selftest_worker(args))
    load_numerics()
    mark("imports done")
    # stdout carries the ONE JSON line and nothing else: RCCL prints a version banner, gloo its connection lines, straight to file
    # descriptor 1 of every rank.  From here on descriptor 1 is stderr; the JSON line goes to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal knob (one-GPU boxes): DRS_BENCH_REHEARSAL=1 runs all ranks on cuda:0 over gloo, to exercise the
    # N > 1 code path without a second device; never set by the driver, and the JSON line says so
    rehearsal = os.environ.get("DRS_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
        # a rehearsal of the LIBRARY-side collectives on one GPU needs a stand-in for RCCL (which refuses two ranks on one device):
        # DRS_BENCH_RCCL_LIB names it (tests/c/nccl_shm_double.cpp built); this script's own variable -- the library reads none, it is
        # told by a call -- and honoured in a rehearsal only
        if os.environ.get("DRS_BENCH_RCCL_LIB"):
            from drs_amd import _lib as _drs_lib
            _drs_lib.call("drs_rccl_bind_library", os.environ["DRS_BENCH_RCCL_LIB"].encode())
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank

    from drs_amd.net import DilatedNet, KernelTimer
    from drs_amd import patches as P
    from drs_amd.synthetic import make_tile, grid_instances
    from drs_amd.dist import TorchComm, shard_slice

    comm = None
    # DRS_FORCE_COLLECTIVES=1 at N = 1: every collective of the step is issued through RCCL anyway (identities on the data): what the
    # collectives' launches and stream hand-overs cost a rank, measurable on a one-GPU box; the JSON line says so
    forced = world == 1 and os.environ.get("DRS_FORCE_COLLECTIVES") == "1"
    if forced:
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533")):
            os.environ.setdefault(k, v)
    t_pg = time.perf_counter()
    if world > 1 or forced:
        comm = TorchComm("gloo" if rehearsal else "nccl")
    t_pg = time.perf_counter() - t_pg
    rank = comm.rank if comm else 0
    mark("process group up")
    B_local = GLOBAL_BATCH // world
    if GLOBAL_BATCH % world:
        sys.exit("global batch %d not divisible by %d ranks" % (GLOBAL_BATCH, world))

    tile, lab = make_tile(TILE, TILE, CHANNELS, CLASSES, seed=1234)
    pool = P.TilePool([tile], [lab], dev, dtype=np.float64)
    inst = grid_instances(TILE, TILE, PATCH, 25, GLOBAL_BATCH * 100, seed=0)      # the super-batch (isprs:1630-1632)
    mean = tile[:, :, :3].mean(axis=(0, 1)).tolist()
    std = tile[:, :, :3].std(axis=(0, 1)).tolist()
    net = DilatedNet(NET, CHANNELS, CLASSES, WD, b_max=B_local, s_max=PATCH, device=dev, seed=42, comm=comm, arith=args.arith)
    sl = shard_slice(GLOBAL_BATCH, rank, world)
    mark("net built, collectives=%s" % (getattr(net, "collectives", None) if comm else None))
    if os.environ.get("DRS_BENCH_ATTEMPT", "default") in os.environ.get("DRS_BENCH_FAKE_HANG", "-").split(","):
        time.sleep(3600)                      # (tests: a hang in the collectives, as the supervisor sees it)

    import random
    random.seed(7)
    np.random.seed(7)
    shuffle = np.asarray(random.sample(range(len(inst)), len(inst)))
    it = 0

    def one_step():
        nonlocal shuffle, it      # `net` is read from the enclosing scope at call time (the opt-in pass rebinds it)
        shuffle, batch, it = P.select_batch(shuffle, GLOBAL_BATCH, it, len(inst))
        rows = inst[batch]
        aug = P.draw_augmentation(rows, PATCH, CHANNELS, noise="device")       # every rank draws the whole batch
        mine = P.Augmentation(B_local)
        mine.rot_on, mine.rot, mine.noise_on, mine.flip, mine.seed, mine.index0 = aug.rot_on[sl], aug.rot[sl], aug.noise_on[sl], aug.flip[sl], aug.seed, sl.start
        P.crop_to_net(net, pool, rows[sl], PATCH, mean, std, mine)
        return net.train_step(B_local, PATCH, LR)      # loss parts, predictions and the confusion matrix stay on the device

    t_first = time.perf_counter()
    one_step()              # with W = 0 still load the code objects / create the communicators outside the timed region
    torch.cuda.synchronize()
    t_first = time.perf_counter() - t_first
    mark("first step done")
    for _ in range(max(0, args.warmup - 1)):
        one_step()
    if comm:
        comm.barrier()
    torch.cuda.synchronize()
    mark("warm-up done")
    # the first contact of this world size with the collectives, per rank: host process-group set-up, seconds inside
    # ncclCommInitRank per library-side communicator, the first all-reduce, the first whole step (RCCL's kernels and connections load there)
    first_contact = None
    if comm:
        mine = dict(rank=rank, process_group_init_s=round(t_pg, 3), first_step_s=round(t_first, 3), **(getattr(net, "first_contact", None) or {}))
        first_contact = dict(per_rank=comm.gather_objects(mine), nccl_debug=os.environ.get("NCCL_DEBUG"),
                             failed_attempt_nccl_tail=os.environ.get("DRS_BENCH_NCCL_TAIL") or None,
                             note="seconds, per rank: torch.distributed process group; ncclCommInitRank per library-side communicator "
                                  "(engine.py _install_rccl); its first all-reduce; the first training step with its collectives")
    # one HIP event per step on the launch stream (an asynchronous record: not on the host's critical path): the MEDIAN step, SURVEY 8(d)
    step_events = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    out = None
    step_events[0].record()
    for k in range(args.steps):
        out = one_step()
        step_events[k + 1].record()
    t_local = time.perf_counter()
    if comm:
        comm.barrier()
    torch.cuda.synchronize()
    elapsed = elapsed_local = time.perf_counter() - t0
    step_ms = sorted(step_events[k].elapsed_time(step_events[k + 1]) for k in range(args.steps))
    per_rank_ms = None
    if comm:
        elapsed = comm.max_float(elapsed, dev)
        mine_ms = 1e3 * elapsed_local / args.steps
        per_rank_ms = dict(min=round(-comm.max_float(-mine_ms, dev), 3), max=round(comm.max_float(mine_ms, dev), 3),
                           note="wall time of the timed region / steps on each rank (barrier included)")
    del t_local
    mark("timed region done")
    loss = net.loss_value(out["loss_parts"])

    # ---- per-kernel timing pass (outside the timed region): HIP events around every launch
    kernels, roofline = None, None
    if not args.no_kernel_timing:
        net.timer = KernelTimer()
        for _ in range(3):
            one_step()
        summ = net.timer.summary()
        net.timer = None
        kernels = {}
        for kind, d in sorted(summ.items()):
            avg_ms = d["ms"] / d["launches"]
            if kind.startswith("allreduce_"):
                # the step's collectives, timed on the stream each is issued on (engine.hip K_AR_*): `work` = bytes.  At N = 1 they only
                # exist with DRS_FORCE_COLLECTIVES=1 (identities: what issuing them costs a rank); at N > 1 this is the wire + latency
                # time the ranks really paid, the figure that says where a scaling loss went
                per_step = d["launches"] // 3
                kernels[kind] = dict(bound="xgmi", calls_per_step=per_step, avg_us=round(1e3 * avg_ms, 2), ms_per_step=round(d["ms"] / 3, 4),
                                     bytes_per_call=int(d["work"] / d["launches"]),
                                     algbw_gbs=round(d["work"] / (d["ms"] * 1e-3) / 1e9, 2))
            elif kind.startswith("conv_"):
                ach = d["work"] / (d["ms"] * 1e-3) / 1e12
                kernels[kind] = dict(bound="mfma", launches_per_step=d["launches"] // 3, avg_ms=round(avg_ms, 4),
                                     achieved=round(ach, 2), peak=ar["peak"], unit="TFLOP/s", frac=round(ach / ar["peak"], 4))
            else:
                ach = d["work"] / (d["ms"] * 1e-3) / 1e9
                kernels[kind] = dict(bound="hbm", launches_per_step=d["launches"] // 3, avg_ms=round(avg_ms, 4),
                                     achieved=round(ach, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(ach / PEAK_HBM_GBS, 4))
        # dominant kernel: conv_igemm_kernel (forward and input-gradient launches are the same kernel)
        w = sum(summ[k]["work"] for k in ("conv_fwd", "conv_dgrad") if k in summ)
        ms = sum(summ[k]["ms"] for k in ("conv_fwd", "conv_dgrad") if k in summ)
        nl = sum(summ[k]["launches"] for k in ("conv_fwd", "conv_dgrad") if k in summ)
        ach = w / (ms * 1e-3) / 1e12
        ex = executed_fraction(net.plan, B_local, PATCH)
        roofline = dict(kernel=ar["kernel"], bound="mfma", achieved=round(ach, 2), peak=ar["peak"],
                        unit="TFLOP/s", frac=round(ach / ar["peak"], 4), traffic=pmc_traffic(args.arith),
                        launches=nl, avg_launch_ms=round(ms / nl, 4), algorithmic_gflop_per_launch=round(w / nl / 1e9, 2),
                        # `frac` prices the ALGORITHMIC flops (every filter tap); taps that meet only the zero halo for a whole tile
                        # are not multiplied, so the MFMA pipe issues `executed_share` of them: executed_frac = frac * executed_share
                        executed_share=round(ex, 4), executed_frac=round(ach * ex / ar["peak"], 4),
                        traffic_source="separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed under "
                                       "profiles/ (tools/pmc_summary.py); not measurable inside this process")
        if ar["products"] > 1:      # the MFMA pipe executes `products` partial products per algorithmic multiply
            roofline["mfma_products_per_multiply"] = ar["products"]
            roofline["mfma_issued_tflops"] = round(ach * ar["products"], 1)
            roofline["mfma_issued_frac"] = round(ach * ar["products"] / ar["peak"], 4)

    mark("kernel timing done")
    # ---- validation half of the metric: forward-only pixels/sec (eval-mode BN, arg-max, confusion), isprs:1569-1618
    vb = 25             # B * 100 patches per validation pass; 25 batches of 128 bound the sample (0.4 s)
    for _ in range(2):
        P.crop_to_net(net, pool, inst[:GLOBAL_BATCH][sl], PATCH, mean, std)
        net.forward(B_local, PATCH, want_logits=False, labels=True)
    if comm:
        comm.barrier()
    torch.cuda.synchronize()
    tv = time.perf_counter()
    for i in range(vb):
        P.crop_to_net(net, pool, inst[i * GLOBAL_BATCH:(i + 1) * GLOBAL_BATCH][sl], PATCH, mean, std)
        net.forward(B_local, PATCH, want_logits=False, labels=True)
    if comm:
        comm.barrier()
    torch.cuda.synchronize()
    tv = time.perf_counter() - tv
    if comm:
        tv = comm.max_float(tv, dev)
    val_pixels_per_s = vb * GLOBAL_BATCH * PATCH * PATCH / tv
    mark("validation done")

    # ---- the opt-in arithmetics beside the headline (N = 1 only; never `value`): the same step on the split-bf16 kernels (op-level path).
    # bf16x6 is the fp32-EQUIVALENT one (tests/test_gpu_split.py: its errors against the fp64 oracle are no larger than the fp32 MFMA
    # kernels' on every BASELINE shape); bf16x3 trades 2^-16 product error for speed.
    opt_in = None
    if world == 1 and args.arith == "f32" and args.opt_in and not args.no_opt_in:
        opt_in = {}
        main_net = net
        for arith in ("bf16x6", "bf16x3"):
            net = DilatedNet(NET, CHANNELS, CLASSES, WD, b_max=B_local, s_max=PATCH, device=dev, seed=42, arith=arith)
            try:
                for _ in range(3):
                    one_step()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                for _ in range(10):
                    one_step()
                torch.cuda.synchronize()
                dt2 = (time.perf_counter() - t2) / 10
            finally:
                del net
                net = main_net
                torch.cuda.empty_cache()
            opt_in[arith] = {"dtype": ARITH[arith]["dtype"], "value": round(GLOBAL_BATCH / dt2, 2), "unit": "patches/s",
                             "ms_per_step": round(1e3 * dt2, 3), "steps": 10, "warmup": 3}
        opt_in["note"] = ("not the headline and not credited: opt-in arithmetics of the convolutions on the op-level path only (DESIGN.md 3a). "
                          "bf16x6 matches the fp32 MFMA kernels on Gaussian operands but its 6-of-9 partial products carry ~2^-22 per "
                          "product, ~3x an fp32 FMA's rounding, when a few large terms dominate a sum "
                          "(tests/test_gpu_split.py::test_three_term_arithmetic_on_adversarial_operands); bf16x3 carries 2^-16")

    mark("opt-in done")
    size_table = None
    if world == 1 and args.arith == "f32" and not args.no_size_table and not forced:
        size_table = per_rank_size_table(dev, pool, mean, std)
    mark("size table done")
    configs = None
    if world == 1 and args.arith == "f32" and not args.no_configs and not forced:
        configs = baseline_configs(dev)
    mark("configs done")

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        value = GLOBAL_BATCH * args.steps / elapsed
        # fp32 ceiling of the whole step: 3 x 17.12 GFLOP per 64 x 64 patch at 157.3 TFLOP/s per GPU (3062 patches/s)
        ceiling = world * PEAK_FP32_MFMA_TFLOPS * 1e12 / (3 * 2 * net.plan.mac_per_pixel() * PATCH * PATCH)
        if roofline is not None and args.arith == "f32":
            roofline["whole_step_frac"] = round(value / ceiling, 4)
            roofline["whole_step_ceiling_patches_per_s"] = round(ceiling, 1)
        line = {
            "metric": "training patches/sec dilated_grsl_rate8 64x64x5", "value": round(value, 2), "unit": "patches/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "median_step_ms": round(step_ms[len(step_ms) // 2], 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": ar["dtype"], "data": "synthetic",
            "config": {"workload": "dilated_grsl_rate8 (Dilated8Pooling) training step, single_fixed 64x64, 5-band synthetic "
                                   "2048x2048 tile, global batch 128 (crop+augment+normalise, fwd, loss, bwd, momentum, confusion)",
                       "global_batch": GLOBAL_BATCH, "patch": PATCH, "bands": CHANNELS, "classes": CLASSES,
                       "parallelism": "dp%d" % world + (" (REHEARSAL: all ranks on one GPU over gloo)" if rehearsal else "")
                                      + (" (collectives forced through RCCL at world 1)" if forced else ""),
                       "sync_bn": True, "collectives": collectives_label(getattr(net, "collectives", None)) if comm else None,
                       "fallback_reason": os.environ.get("DRS_BENCH_FALLBACK_REASON") or None,
                       # the sum of one `1` per rank over the step's own communicator (engine.py: the known-answer check of a new
                       # RCCL communicator, or the same through the callback's): > 1 only if that many ranks really took part
                       "rccl_ranks_observed": getattr(net, "ranks_observed", 1) if comm else 1},
            "val_pixels_per_sec": round(val_pixels_per_s, 1), "final_loss": round(loss, 5),
            "workspace_gb_per_gpu": round(net.workspace_bytes() / 1e9, 2),
            "train_tflops": round(value * 3 * 2 * net.plan.mac_per_pixel() * PATCH * PATCH / 1e12, 2),
            "roofline": roofline, "kernels": kernels, "cpu_baseline": cpu, "opt_in_arithmetic": opt_in,
            "extra": {"configs": configs, "first_contact": first_contact, "per_rank_size_table": size_table, "per_rank_ms": per_rank_ms,
                      "step_ms_events": dict(median=round(step_ms[len(step_ms) // 2], 3), min=round(step_ms[0], 3), max=round(step_ms[-1], 3),
                                             note="HIP events between consecutive steps on rank 0's launch stream; `value` stays steps / wall time")},
        }
        if size_table and size_table.get("headline_shape"):
            hs = size_table["headline_shape"]
            hs["ideal_speedup_before_wire"] = round(line["ms_per_step"] / hs["ms_per_step"], 3)
            hs["note"] = ("one rank's step of the 8-GPU run of this headline (16 patches of 64 x 64, no collectives) against the 1-GPU step "
                          "above: what 8 ranks could reach before any wire time")
        if world > 1 or forced:
            line["extra"]["per_rank_kernels_note"] = ("`kernels` are rank 0's families at the per-rank batch %d, the allreduce_* rows its "
                                                      "collectives (HIP events on the stream each is issued on)" % B_local)
        os.write(json_fd, (json.dumps(line) + "\n").encode())
        mark("json written")
    if comm:
        torch.cuda.synchronize()
        comm.barrier()
        if hasattr(net, "close"):
            net.close()              # the library's own RCCL communicators go before the process group does
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
