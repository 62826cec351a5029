"""CPU: the launch plumbing of bench.py for N > 1 (no GPU: DRS_BENCH_SELFTEST=1 replaces the measurement by a gloo barrier).

`python bench.py --gpus N` without a launcher must start torch.distributed.run itself as a fresh child; under either launch shape
every rank is a supervisor that starts the measuring process as a child, watches its progress markers and, on a hang or a crash,
kills it and starts a FRESH child on the next collectives path (DRS_COMM=torch, then DRS_RCCL_ASYNC=1) -- the one 8-GPU run
a round gets must not be lost to a hang in the dual-communicator RCCL path."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(cmd, tmp_path, **env):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "DRS_BENCH_CHILD", "DRS_COMM", "DRS_RCCL_ASYNC"):
        e.pop(k, None)
    e.update(DRS_BENCH_SELFTEST="1", **env)
    r = subprocess.run(cmd, cwd=str(tmp_path), env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    out = r.stdout.decode()
    lines = [ln for ln in out.splitlines() if ln.strip()]
    return r.returncode, lines, r.stderr.decode()


def test_plain_launch_starts_the_launcher_itself(tmp_path):
    rc, lines, err = _run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1"], tmp_path)
    assert rc == 0, err
    assert len(lines) == 1                                   # ONE JSON line on stdout, nothing else
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["config"]["ranks_observed"] == 2
    assert d["config"]["collectives"] == "selftest" and d["config"]["fallback_reason"] is None
    assert "stage=warm-up done" in err
    # r05: a default run that worked is followed by ONE more fresh measuring process per rank with DRS_RCCL_BUCKETS=2; its line rides
    # inside the first one and never replaces its value
    sp = d["extra"]["second_pass"]
    assert sp["config"]["env"]["DRS_RCCL_BUCKETS"] == "2" and sp["config"]["env"]["DRS_COMM"] == "rccl" and sp["config"]["ranks_observed"] == 2
    assert sp["value"] == 1.0 and d["config"]["env"]["DRS_RCCL_BUCKETS"] is None


def test_a_second_pass_that_dies_or_hangs_leaves_the_first_line_standing(tmp_path):
    rc, lines, err = _run([sys.executable, BENCH, "--gpus", "2"], tmp_path, DRS_BENCH_FAKE_CRASH="buckets")
    assert rc == 0, err
    d = json.loads(lines[-1])
    assert len(lines) == 1 and d["value"] == 1.0 and "exit code 7" in d["extra"]["second_pass"]["error"]
    rc, lines, err = _run([sys.executable, BENCH, "--gpus", "2"], tmp_path, DRS_BENCH_FAKE_HANG="buckets", DRS_BENCH_SECOND_PASS_LIMIT_S="4")
    assert rc == 0, err
    d = json.loads(lines[-1])
    assert len(lines) == 1 and d["value"] == 1.0 and "timeout" in d["extra"]["second_pass"]["error"]
    rc, lines, err = _run([sys.executable, BENCH, "--gpus", "2"], tmp_path, DRS_BENCH_SECOND_PASS="0")
    assert rc == 0 and "extra" not in json.loads(lines[-1])


def test_under_the_launcher_every_rank_supervises_a_fresh_child(tmp_path):
    port = 23000 + os.getpid() % 5000
    rc, lines, err = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                           "--master-port", str(port), BENCH, "--gpus", "2"], tmp_path)
    assert rc == 0, err
    assert len(lines) == 1 and json.loads(lines[0])["config"]["ranks_observed"] == 2


def test_a_hang_before_warm_up_falls_back_to_the_callback_path(tmp_path):
    rc, lines, err = _run([sys.executable, BENCH, "--gpus", "2"], tmp_path, DRS_BENCH_FAKE_HANG="default", DRS_BENCH_WATCHDOG_S="4")
    assert rc == 0, err
    d = json.loads(lines[-1])
    assert len(lines) == 1 and d["config"]["collectives"] == "selftest (fallback after timeout)"
    assert "timeout" in d["config"]["fallback_reason"] and "process group up" in d["config"]["fallback_reason"]
    assert d["config"]["env"]["DRS_COMM"] == "torch" and d["config"]["ranks_observed"] == 2
    assert err.count("killing the process group") == 2      # both ranks' supervisors killed their own child


def test_crashes_walk_the_chain_to_its_last_entry(tmp_path):
    rc, lines, err = _run([sys.executable, BENCH, "--gpus", "2"], tmp_path, DRS_BENCH_FAKE_CRASH="default,torch", DRS_BENCH_WATCHDOG_S="30")
    assert rc == 0, err
    d = json.loads(lines[-1])
    assert d["config"]["collectives"] == "selftest (fallback after failure)"
    assert d["config"]["env"]["DRS_RCCL_ASYNC"] == "1" and d["config"]["env"]["DRS_COMM"] == "rccl"


def test_every_attempt_failing_is_reported_not_hung(tmp_path):
    rc, lines, err = _run([sys.executable, BENCH, "--gpus", "2"], tmp_path, DRS_BENCH_FAKE_CRASH="default,torch,async", DRS_BENCH_WATCHDOG_S="30")
    assert rc != 0
    d = json.loads(lines[-1])
    assert d["value"] is None and "every attempt failed" in d["error"]


def _alive_with_token(token):
    """pids of live processes whose environment carries `token` (every process the launch started inherits it)"""
    pids = []
    for d in os.listdir("/proc"):
        if not d.isdigit() or int(d) == os.getpid():
            continue
        try:
            env = open("/proc/%s/environ" % d, "rb").read()
            stat = open("/proc/%s/stat" % d).read()
        except OSError:
            continue
        if token.encode() in env and ") Z " not in stat:
            pids.append(int(d))
    return pids


@pytest.mark.parametrize("sig", ["TERM", "KILL"])
def test_no_child_outlives_a_parent_that_is_killed_during_a_hang(tmp_path, sig):
    """the measuring processes run in sessions of their own; `timeout ... python bench.py --gpus N`, Ctrl-C or the launcher
    tearing the ranks down must not leave them on the GPUs.  SIGTERM: the handlers take the groups down; SIGKILL: PR_SET_PDEATHSIG does."""
    import signal
    import time
    import uuid
    token = "drs-token-" + uuid.uuid4().hex
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "DRS_BENCH_CHILD", "DRS_COMM", "DRS_RCCL_ASYNC"):
        e.pop(k, None)
    e.update(DRS_BENCH_SELFTEST="1", DRS_BENCH_FAKE_HANG="default", DRS_BENCH_WATCHDOG_S="600", DRS_BENCH_WATCHDOG_STAGE_S="600", DRS_TEST_TOKEN=token)
    errf = open(str(tmp_path / "err.txt"), "wb")
    p = subprocess.Popen([sys.executable, BENCH, "--gpus", "2"], cwd=str(tmp_path), env=e, stdout=subprocess.DEVNULL, stderr=errf)
    try:
        t_end = time.time() + 240
        while time.time() < t_end:                         # both measuring children are up and hanging
            if open(str(tmp_path / "err.txt"), "rb").read().count(b"stage=process group up") >= 2:
                break
            assert p.poll() is None, open(str(tmp_path / "err.txt")).read()
            time.sleep(0.5)
        else:
            raise AssertionError("the children never reached the hang: " + open(str(tmp_path / "err.txt")).read())
        assert len(_alive_with_token(token)) >= 5          # parent, launcher, two supervisors, two measuring processes (minus this one)
        p.send_signal(signal.SIGTERM if sig == "TERM" else signal.SIGKILL)
        p.wait(timeout=60)
        assert p.returncode != 0
        t_end = time.time() + 60
        while time.time() < t_end and _alive_with_token(token):
            time.sleep(0.5)
        left = _alive_with_token(token)
        assert not left, "processes that outlived the parent: %s" % [open("/proc/%d/cmdline" % q, "rb").read() for q in left]
    finally:
        for q in _alive_with_token(token):
            try:
                os.kill(q, signal.SIGKILL)
            except OSError:
                pass
        if p.poll() is None:
            p.kill()


@pytest.mark.parametrize("shape", ["launcher", "plain"])
def test_a_teardown_during_the_second_pass_still_delivers_the_first_line(tmp_path, shape):
    """the finished default measurement used to be written only after the optional second pass returned; a driver
    time-out or the launcher's teardown during that pass lost it.  Now the supervisor holds the line and its signal handler writes it."""
    import signal
    import time
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "DRS_BENCH_CHILD", "DRS_COMM", "DRS_RCCL_ASYNC"):
        e.pop(k, None)
    e.update(DRS_BENCH_SELFTEST="1", DRS_BENCH_FAKE_HANG="buckets", DRS_BENCH_SECOND_PASS_LIMIT_S="600", DRS_BENCH_SECOND_PASS_WALL_S="600")
    port = 24000 + os.getpid() % 5000
    cmd = [sys.executable, BENCH, "--gpus", "2"]
    if shape == "launcher":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
               str(port)] + cmd[1:]
    outf, errf = open(str(tmp_path / "out.txt"), "wb"), open(str(tmp_path / "err.txt"), "wb")
    p = subprocess.Popen(cmd, cwd=str(tmp_path), env=e, stdout=outf, stderr=errf)
    try:
        t_end = time.time() + 240
        while time.time() < t_end:                         # first pass done (2 x warm-up done), second pass up and hanging (4 x process group up)
            err = open(str(tmp_path / "err.txt"), "rb").read()
            if err.count(b"stage=process group up") >= 4:
                break
            assert p.poll() is None, err.decode()
            time.sleep(0.5)
        else:
            raise AssertionError("the second pass never reached its hang: " + open(str(tmp_path / "err.txt")).read())
        assert not open(str(tmp_path / "out.txt"), "rb").read().strip()      # nothing is out yet: the line is waiting for the second pass
        p.send_signal(signal.SIGTERM)
        p.wait(timeout=90)
    finally:
        if p.poll() is None:
            p.kill()
    lines = [ln for ln in open(str(tmp_path / "out.txt")).read().splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, (lines, open(str(tmp_path / "err.txt")).read()[-2000:])
    d = json.loads(lines[0])
    assert d["value"] == 1.0 and d["n_gpus"] == 2 and "second_pass" not in (d.get("extra") or {})


def test_the_second_pass_has_an_overall_wall_clock_cap(tmp_path):
    """markers keep coming but the pass never ends (here: the cap is shorter than the pass): cut by wall time, first line intact"""
    rc, lines, err = _run([sys.executable, BENCH, "--gpus", "2"], tmp_path, DRS_BENCH_SECOND_PASS_WALL_S="0.5")
    assert rc == 0, err
    d = json.loads(lines[-1])
    assert len(lines) == 1 and d["value"] == 1.0
    sp = d["extra"]["second_pass"]
    assert "timeout" in sp.get("error", "") or sp.get("value") == 1.0       # (a pass quicker than the cap's one-second poll is allowed to finish)
