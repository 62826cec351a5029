"""-m gpu: the data-parallel training step of the HIP path with 2 ranks (two processes sharing the one GPU of the
test box, gloo as the transport so that no second device is needed): sharded batch + sync-BN sums + gradient
all-reduce must reproduce the single-rank step on the full batch."""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

NET, CH, K, B, S = "dilated8_grsl", 5, 6, 4, 21


def _inputs():
    rng = np.random.default_rng(0)
    return rng.normal(size=(B, S * S * CH)).astype(np.float32), rng.integers(0, K, size=(B, S * S))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from drs_amd.dist import TorchComm, shard_slice
    from drs_amd.net import DilatedNet
    torch.cuda.set_device(0)
    comm = TorchComm("gloo")
    x, y = _inputs()
    sl = shard_slice(B, rank, world)
    d = DilatedNet(NET, CH, K, 0.005, b_max=B // world, s_max=S, device="cuda:0", seed=3, comm=comm)
    d.feed(x[sl], y[sl], S)
    res = d.train_step(B // world, S, 0.01)
    torch.cuda.synchronize()
    if rank == 0:
        np.savez(out, grads=d.grads.cpu().numpy(), params=d.params.cpu().numpy(), bn=d.bn.cpu().numpy(),
                 loss=d.loss_value(res["loss_parts"]), conf=res["conf"].cpu().numpy())
    comm.barrier()
    dist.destroy_process_group()


def test_two_rank_step_equals_single_rank(tmp_path):
    from drs_amd.net import DilatedNet
    out = str(tmp_path / "dp.npz")
    mp.spawn(_worker, args=(2, 29600 + os.getpid() % 1000, out), nprocs=2, join=True)
    x, y = _inputs()
    d = DilatedNet(NET, CH, K, 0.005, b_max=B, s_max=S, device="cuda:0", seed=3)
    d.feed(x, y, S)
    res = d.train_step(B, S, 0.01)
    torch.cuda.synchronize()
    r = np.load(out)

    def rel(a, b):
        return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))
    assert abs(float(r["loss"]) - d.loss_value(res["loss_parts"])) < 1e-6
    # same sums in a different grouping: the last-bit differences in the BN statistics flip a few ReLU signs /
    # pool winners, so gradients agree to ~1e-4, not to rounding
    assert rel(r["grads"], d.grads.cpu().numpy()) < 2e-3
    assert rel(r["params"], d.params.cpu().numpy()) < 1e-4
    assert rel(r["bn"], d.bn.cpu().numpy()) < 1e-6
    np.testing.assert_array_equal(r["conf"], res["conf"].cpu().numpy())


# ---------------------------------------------------------------------------------------------------------------------
# several steps, decision-aligned: data-parallel equivalence beyond one step.  Two correct runs drift apart step by step through
# ReLU signs / pool winners that flip on last-bit differences of the batch statistics, so a free comparison of a
# two-rank run with a one-rank run says little after the first update; the ORACLE can be told the ranks' own discrete decisions
# (concatenated along the batch) and then has to reproduce every continuous quantity of every step tightly -- the same device the
# single-rank trajectory tests use (tests/test_gpu_net.py).
DP_STEPS = 4


def _dp_inputs(step):
    rng = np.random.default_rng(100 + step)
    return rng.normal(size=(B, S, S, CH)).astype(np.float32), rng.integers(0, K, size=(B, S, S))


def _multi_worker(rank, world, port, root):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from drs_amd.dist import TorchComm, shard_slice
    from drs_amd.net import DilatedNet
    torch.cuda.set_device(0)
    comm = TorchComm("gloo")
    sl = shard_slice(B, rank, world)
    b = B // world
    d = DilatedNet(NET, CH, K, 0.005, b_max=b, s_max=S, device="cuda:0", seed=3, comm=comm)
    M = b * S * S
    for step in range(DP_STEPS):
        x, y = _dp_inputs(step)
        d.feed(x[sl].reshape(b, -1), y[sl].reshape(b, -1), S)
        res = d.train_step(b, S, 0.01)
        torch.cuda.synchronize()
        dec = {}
        for i, L in enumerate(d.plan.layers):
            z = d.z[i][:M * L.cout].cpu().numpy().reshape(b, S, S, L.cout)
            mr = d.mean_rstd[i].cpu().numpy().reshape(L.cout, 2)
            dec["pos%d" % i] = (z - mr[:, 0]) * mr[:, 1] > 0
            dec["idx%d" % i] = d.idx[i][:M * L.cout].cpu().numpy().reshape(b, S, S, L.cout)
        np.savez(os.path.join(root, "dec_r%d_s%d.npz" % (rank, step)), loss=d.loss_value(res["loss_parts"]), conf=res["conf"].cpu().numpy(), **dec)
    if rank == 0:
        np.savez(os.path.join(root, "final.npz"), **d.state_dict())
    comm.barrier()
    dist.destroy_process_group()


def test_two_rank_trajectory_follows_the_decision_aligned_oracle(tmp_path):
    from oracle import tf_ops as T
    from drs_amd.net import DilatedNet
    root = str(tmp_path)
    mp.spawn(_multi_worker, args=(2, 29800 + os.getpid() % 1000, root), nprocs=2, join=True)
    ref = DilatedNet(NET, CH, K, 0.005, b_max=B, s_max=S, device="cuda:0", seed=3)      # the same initial variables (seed 3)
    o = T.OracleNet(NET, CH, K, dtype=np.float64, seed=0)
    for n in ref.variable_names():
        o.p[n] = ref.get_variable(n).astype(np.float64)
    nl = len(ref.plan.layers)
    for step in range(DP_STEPS):
        x, y = _dp_inputs(step)
        r = [np.load(os.path.join(root, "dec_r%d_s%d.npz" % (q, step))) for q in range(2)]
        dec = [{"pos": np.concatenate([r[0]["pos%d" % i], r[1]["pos%d" % i]]), "idx": np.concatenate([r[0]["idx%d" % i], r[1]["idx%d" % i]])}
               for i in range(nl)]
        lo, _ = o.train_step(x.astype(np.float64), y, 0.01, 0.005, decisions=dec)
        assert float(r[0]["loss"]) == float(r[1]["loss"])                                # every rank holds the global loss
        assert abs(float(r[0]["loss"]) - lo) < 1e-4 * abs(lo), (step, float(r[0]["loss"]), lo)
        np.testing.assert_array_equal(r[0]["conf"], r[1]["conf"])
        assert int(r[0]["conf"].sum()) == B * S * S                                      # the confusion matrix counts the GLOBAL batch
    fin = np.load(os.path.join(root, "final.npz"))

    def rel(a, b):
        return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))
    for n in ref.plan.offsets:
        assert rel(fin[n], o.p[n]) < 1e-4, n
        if not (n.endswith("/biases") and n != "conv_classifier/biases"):
            assert rel(fin[n + "/Momentum"], o.mom[n]) < 1e-3, n
    for n in ref.variable_names():
        if "moving" in n:
            assert rel(fin[n], o.p[n]) < 1e-5, n
    assert int(fin["main_global_step"]) == DP_STEPS


# one step at a REAL per-rank shape of the 8-GPU run: 16 patches per rank, a side that takes the stream-K
# forward launches and the table form of the filter gradient, sync-BN through the callback with the finish folded into the
# normalising kernel -- against the decision-aligned oracle on the global batch.
RS_B, RS_S = 16, 55


def _real_shape_inputs():
    rng = np.random.default_rng(77)
    return rng.normal(size=(2 * RS_B, RS_S, RS_S, CH)).astype(np.float32), rng.integers(0, K, size=(2 * RS_B, RS_S, RS_S))


def _real_shape_worker(rank, world, port, root):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from drs_amd.dist import TorchComm, shard_slice
    from drs_amd.net import DilatedNet
    torch.cuda.set_device(0)
    comm = TorchComm("gloo")
    x, y = _real_shape_inputs()
    sl = shard_slice(2 * RS_B, rank, world)
    d = DilatedNet(NET, CH, K, 0.005, b_max=RS_B, s_max=RS_S, device="cuda:0", seed=3, comm=comm)
    M = RS_B * RS_S * RS_S
    d.feed(x[sl].reshape(RS_B, -1), y[sl].reshape(RS_B, -1), RS_S)
    res = d.train_step(RS_B, RS_S, 0.01)
    torch.cuda.synchronize()
    dec = {}
    for i, L in enumerate(d.plan.layers):
        z = d.z[i][:M * L.cout].cpu().numpy().reshape(RS_B, RS_S, RS_S, L.cout)
        mr = d.mean_rstd[i].cpu().numpy().reshape(L.cout, 2)
        dec["pos%d" % i] = (z - mr[:, 0]) * mr[:, 1] > 0
        dec["idx%d" % i] = d.idx[i][:M * L.cout].cpu().numpy().reshape(RS_B, RS_S, RS_S, L.cout)
    np.savez(os.path.join(root, "rs_r%d.npz" % rank), loss=d.loss_value(res["loss_parts"]), conf=res["conf"].cpu().numpy(), **dec)
    if rank == 0:
        np.savez(os.path.join(root, "rs_final.npz"), **d.state_dict())
    comm.barrier()
    dist.destroy_process_group()


def test_two_ranks_at_a_real_per_rank_shape_follow_the_decision_aligned_oracle(tmp_path):
    from oracle import tf_ops as T
    from drs_amd import _lib
    from drs_amd.net import DilatedNet
    # the shape does take the paths it is here for (the product library's own rules)
    assert RS_S % 32 != 0
    assert _lib.query("drs_conv_halo_skip", RS_B, RS_S, 3, 8, 8, 256, 256) == 0          # conv8 forward: a stream-K launch (no halo skipping there)
    root = str(tmp_path)
    mp.spawn(_real_shape_worker, args=(2, 29850 + os.getpid() % 1000, root), nprocs=2, join=True)
    ref = DilatedNet(NET, CH, K, 0.005, b_max=1, s_max=8, device="cuda:0", seed=3)        # the same initial variables (seed 3)
    o = T.OracleNet(NET, CH, K, dtype=np.float64, seed=0)
    for n in ref.variable_names():
        o.p[n] = ref.get_variable(n).astype(np.float64)
    x, y = _real_shape_inputs()
    r = [np.load(os.path.join(root, "rs_r%d.npz" % q)) for q in range(2)]
    nl = len(ref.plan.layers)
    dec = [{"pos": np.concatenate([r[0]["pos%d" % i], r[1]["pos%d" % i]]), "idx": np.concatenate([r[0]["idx%d" % i], r[1]["idx%d" % i]])} for i in range(nl)]
    lo, _ = o.train_step(x.astype(np.float64), y, 0.01, 0.005, decisions=dec)
    assert float(r[0]["loss"]) == float(r[1]["loss"])
    assert abs(float(r[0]["loss"]) - lo) < 1e-4 * abs(lo), (float(r[0]["loss"]), lo)
    np.testing.assert_array_equal(r[0]["conf"], r[1]["conf"])
    assert int(r[0]["conf"].sum()) == 2 * RS_B * RS_S * RS_S
    fin = np.load(os.path.join(root, "rs_final.npz"))

    def rel(a, b):
        return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))
    for n in ref.plan.offsets:
        assert rel(fin[n], o.p[n]) < 1e-4, n
    for n in ref.variable_names():
        if "moving" in n:
            assert rel(fin[n], o.p[n]) < 1e-5, n


def _build_latency_double(tmp):
    """tools/ubench/nccl_latency_double.hip: a world-1 stand-in for RCCL whose all-reduces are REAL launches on the stream they are given
    (real RCCL launches nothing for a sum over one rank), each holding its stream for a wire time"""
    import subprocess
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "ubench", "nccl_latency_double.hip")
    out = os.path.join(str(tmp), "libnccl_latency_double.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", src, "-o", out], check=True)
    return out


def _rccl_worker(rank, world, port, out, mode):
    """ONE rank on the real backend ('nccl' = RCCL) with every collective of the step forced on (sums over one rank are identities):
    communicator bound to the device, asynchronous gradient buckets, sync-BN sums, stream waits -- the code an 8-GPU run executes."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", DRS_FORCE_COLLECTIVES="1",
                      DRS_COMM={"rccl": "rccl", "launches": "rccl", "callback": "torch", "op": "torch"}[mode])
    if mode == "launches":      # every sum a real launch with 20 us of wire time in the stream it is issued on (the library binds the named stand-in)
        os.environ.update(NCCL_DOUBLE_ALPHA_US="20", NCCL_DOUBLE_GBS="100")
        from drs_amd import _lib
        _lib.call("drs_rccl_bind_library", os.path.join(os.path.dirname(out), "libnccl_latency_double.so").encode())
    engine = mode != "op"
    import torch.distributed as dist
    from drs_amd.dist import TorchComm
    from drs_amd.net import DilatedNet
    torch.cuda.set_device(0)
    comm = TorchComm("nccl")
    assert comm.collective and comm.world == 1 and dist.get_backend() == "nccl"
    x, y = _inputs()
    d = DilatedNet(NET, CH, K, 0.005, b_max=B, s_max=S, device="cuda:0", seed=3, comm=comm, engine=engine)
    calls = []
    real = comm.all_reduce_sum_async
    comm.all_reduce_sum_async = lambda t: (calls.append(t.numel()), real(t))[1]
    losses = []
    for _ in range(3):
        d.feed(x, y, S)
        res = d.train_step(B, S, 0.01)
        losses.append(d.loss_value(res["loss_parts"]))
    torch.cuda.synchronize()
    if mode in ("rccl", "launches"):       # the library issued every sum itself (drs_net_set_rccl): nothing came back into Python
        assert d.collectives.startswith("rccl") and not calls
    else:
        assert len(calls) >= 3 * 3, calls                   # gradient buckets and backward BN sums went through the communicator
    np.savez(out, grads=d.grads.cpu().numpy(), params=d.params.cpu().numpy(), bn=d.bn.cpu().numpy(), losses=np.asarray(losses),
             conf=res["conf"].cpu().numpy())
    comm.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["rccl", "launches", "callback", "op"],
                         ids=["step-level-library-rccl", "step-level-library-sums-as-real-launches", "step-level-callback", "op-level"])
def test_rccl_collectives_at_world_one_leave_the_step_unchanged(tmp_path, mode):
    from drs_amd.net import DilatedNet
    engine = mode != "op"
    out = str(tmp_path / "rccl.npz")
    if mode == "launches":
        _build_latency_double(tmp_path)
    mp.spawn(_rccl_worker, args=(1, 29700 + os.getpid() % 1000, out, mode), nprocs=1, join=True)
    x, y = _inputs()
    d = DilatedNet(NET, CH, K, 0.005, b_max=B, s_max=S, device="cuda:0", seed=3, engine=engine)
    losses = []
    for _ in range(3):
        d.feed(x, y, S)
        res = d.train_step(B, S, 0.01)
        losses.append(d.loss_value(res["loss_parts"]))
    torch.cuda.synchronize()
    r = np.load(out)
    # identities on the data; the only difference allowed is the multi-rank form of the batch-norm statistics (tile sums -> fp64
    # sums -> all-reduce -> finish, against the fused finish): same arithmetic, so the three steps must agree to rounding
    np.testing.assert_allclose(r["losses"], losses, rtol=1e-6)
    np.testing.assert_allclose(r["params"], d.params.cpu().numpy(), rtol=0, atol=1e-6 * float(d.params.abs().max()))
    np.testing.assert_allclose(r["bn"], d.bn.cpu().numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_array_equal(r["conf"], res["conf"].cpu().numpy())
    print("RCCL world-1 run vs no communicator: params bitwise equal = %s" % np.array_equal(r["params"], d.params.cpu().numpy()))


def _rccl_forms_worker(rank, world, port, out, lib=None):
    """every form of the library-side collectives (drs_rccl_form: inline, asynchronous, inline + two overlapped gradient buckets) x the
    one- and two-stream backward pass, in ONE process on the real backend at world 1 with every collective forced on"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", DRS_FORCE_COLLECTIVES="1", DRS_COMM="rccl")
    from drs_amd import _lib
    if lib:      # every sum a real launch that holds its stream for 30 us and keeps NaNs in the buffer meanwhile (nccl_latency_double.hip)
        os.environ.update(NCCL_DOUBLE_ALPHA_US="30", NCCL_DOUBLE_GBS="50", NCCL_DOUBLE_POISON="1")
        _lib.call("drs_rccl_bind_library", lib.encode())
        assert _lib.load().drs_rccl_bind_library(lib.encode()) == 1       # once bound, a second naming is rejected
    import torch.distributed as dist
    from drs_amd.dist import TorchComm
    from drs_amd.net import DilatedNet
    torch.cuda.set_device(0)
    comm = TorchComm("nccl")
    x, y = _inputs()
    res = {}
    for form, env in (("inline", {}), ("async", {"DRS_RCCL_ASYNC": "1"}), ("buckets", {"DRS_RCCL_BUCKETS": "2"}), ("async_word", {"DRS_RCCL_ASYNC": "yes"})):
        for two in ("0", "1"):
            for k in ("DRS_RCCL_ASYNC", "DRS_RCCL_BUCKETS"):
                os.environ.pop(k, None)
            os.environ.update(env)
            want = {"inline": 1, "async": 2, "buckets": 3, "async_word": 1}[form]        # atoi("yes") == 0: the LIBRARY's reading, and the label follows it
            assert _lib.query("drs_rccl_form") == want
            d = DilatedNet(NET, CH, K, 0.005, b_max=B, s_max=S, device="cuda:0", seed=3, comm=comm)
            d.set_two_streams(int(two))
            assert d.collectives.startswith({1: "rccl (inline:", 2: "rccl (asynchronous", 3: "rccl (inline + two overlapped"}[want]), d.collectives
            for _ in range(3):
                d.feed(x, y, S)
                r = d.train_step(B, S, 0.01)
            torch.cuda.synchronize()
            res["%s_%s_params" % (form, two)] = d.params.cpu().numpy()
            res["%s_%s_bn" % (form, two)] = d.bn.cpu().numpy()
            res["%s_%s_conf" % (form, two)] = r["conf"].cpu().numpy()
            res["%s_%s_loss" % (form, two)] = np.asarray(d.loss_value(r["loss_parts"]))
            d.close()
    np.savez(out, **res)
    comm.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("transport", ["rccl", "poisoned-in-flight"])
def test_every_form_of_the_library_side_collectives_gives_the_inline_forms_bits(tmp_path, transport):
    """the asynchronous form is reachable only through DRS_RCCL_ASYNC and no test set it; r05 adds DRS_RCCL_BUCKETS.  Sums over
    one rank are identities and every form issues the same kernels on the same operands, so variables, moving statistics, loss and
    confusion matrix after three steps must be the inline form's bit for bit -- with the one- and the two-stream backward pass.
    Real RCCL launches nothing for a sum over one rank, so with it a missing cross-stream hand-over could not show; the second
    transport (tools/ubench/nccl_latency_double.hip, NCCL_DOUBLE_POISON) makes every sum three launches on the stream it is given --
    save the buffer and fill it with NaNs, 30 us + of wire time, restore: a consumer not ordered behind the collective reads NaNs, a
    producer not ordered in front of it is overwritten by the stale copy.  The one-stream inline form has no hand-over to miss."""
    out = str(tmp_path / "forms.npz")
    lib = _build_latency_double(tmp_path) if transport != "rccl" else None
    mp.spawn(_rccl_forms_worker, args=(1, 29750 + os.getpid() % 1000, out, lib), nprocs=1, join=True)
    r = np.load(out)
    assert np.all(np.isfinite(r["inline_0_params"])) and np.all(np.isfinite(r["inline_0_loss"]))
    for form in ("inline", "async", "buckets", "async_word"):
        for two in ("0", "1"):
            for what in ("params", "bn", "conf", "loss"):
                np.testing.assert_array_equal(r["%s_%s_%s" % (form, two, what)], r["inline_0_%s" % what], err_msg="%s two_streams=%s %s" % (form, two, what))


# ---------------------------------------------------------------------------------------------------------------------
# The library-side collectives at world > 1 on a one-GPU box.  RCCL refuses two ranks on one device, so until r05 the code that an
# 8-GPU run executes -- engine.hip issuing the step's sums itself (inline / two buckets / asynchronous), beside the two-stream
# backward pass -- had only ever run at world 1 (identities).  tests/c/nccl_shm_double.cpp is a shared-memory stand-in for the five
# NCCL entry points the library binds (named through drs_rccl_bind_library); the host group (gloo) only carries the communicator ids.  At two ranks a sum
# of two operands is the same in any order, so every form must give the callback path's bits.
DBL_B, DBL_S = 8, 33           # per rank: stream-K forward launches, the row-segment filter gradient, the two-stream backward pass


def _build_nccl_double(tmp):
    import subprocess
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c", "nccl_shm_double.cpp")
    out = os.path.join(str(tmp), "libnccl_shm_double.so")
    subprocess.run(["g++", "-O1", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", src, "-L/opt/rocm/lib", "-lamdhip64", "-lrt",
                    "-o", out], check=True)
    return out


def _double_inputs(world):
    rng = np.random.default_rng(5)
    return rng.normal(size=(world * DBL_B, DBL_S * DBL_S * CH)).astype(np.float32), rng.integers(0, K, size=(world * DBL_B, DBL_S * DBL_S))


def _double_worker(rank, world, port, out, lib, cases):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), DRS_RCCL_INIT_TIMEOUT_S="150")
    import torch.distributed as dist
    from drs_amd import _lib
    from drs_amd.dist import TorchComm, shard_slice
    _lib.call("drs_rccl_bind_library", lib.encode())       # the shared-memory stand-in for the five NCCL entry points the library binds
    from drs_amd.net import DilatedNet
    torch.cuda.set_device(0)
    comm = TorchComm("gloo")
    x, y = _double_inputs(world)
    sl = shard_slice(world * DBL_B, rank, world)
    res = {}
    for name, env in cases:
        for k in ("DRS_COMM", "DRS_RCCL_ASYNC", "DRS_RCCL_BUCKETS"):
            os.environ.pop(k, None)
        env = dict(env)
        two = env.pop("two_streams", None)
        os.environ.update(env)
        d = DilatedNet(NET, CH, K, 0.005, b_max=DBL_B, s_max=DBL_S, device="cuda:0", seed=3, comm=comm)
        d.set_two_streams(two)
        if env.get("DRS_COMM") == "rccl":
            assert d.collectives.startswith("rccl") and d.ranks_observed == world, (d.collectives, getattr(d, "ranks_observed", None))
        else:
            assert d.collectives == "callback", d.collectives
        losses = []
        for _ in range(3):
            d.feed(x[sl], y[sl], DBL_S)
            r = d.train_step(DBL_B, DBL_S, 0.01)
            losses.append(d.loss_value(r["loss_parts"]))
        torch.cuda.synchronize()
        res[name + "_params"] = d.params.cpu().numpy()
        res[name + "_bn"] = d.bn.cpu().numpy()
        res[name + "_conf"] = r["conf"].cpu().numpy()
        res[name + "_loss"] = np.asarray(losses)
        d.close()
        comm.barrier()
    if rank == 0:
        np.savez(out, **res)
    comm.barrier()
    dist.destroy_process_group()


def test_library_side_collectives_at_two_ranks_give_the_callback_paths_bits(tmp_path):
    lib = _build_nccl_double(tmp_path)
    rccl = {"DRS_COMM": "rccl"}
    cases = [("callback", {"DRS_COMM": "torch"})]
    for form, env in (("inline", {}), ("buckets", {"DRS_RCCL_BUCKETS": "2"}), ("async", {"DRS_RCCL_ASYNC": "1"})):
        for two in ("0", "1"):
            cases.append(("%s_%s" % (form, two), dict(rccl, two_streams=int(two), **env)))
    out = str(tmp_path / "double.npz")
    mp.spawn(_double_worker, args=(2, 29850 + os.getpid() % 1000, out, lib, cases), nprocs=2, join=True)
    r = np.load(out)
    assert np.all(np.isfinite(r["callback_loss"])) and r["callback_loss"][2] != r["callback_loss"][0]
    for name, _ in cases[1:]:
        for what in ("params", "bn", "conf", "loss"):
            np.testing.assert_array_equal(r["%s_%s" % (name, what)], r["callback_%s" % what], err_msg="%s %s" % (name, what))


def test_library_side_collectives_at_three_ranks_follow_the_callback_path(tmp_path):
    """three operands: the stand-in adds in rank order, gloo in its own, so the sums differ in the last bit and a few ReLU signs /
    pool winners flip (the bounds are those of test_two_rank_step_equals_single_rank)"""
    lib = _build_nccl_double(tmp_path)
    cases = [("callback", {"DRS_COMM": "torch"}), ("inline", {"DRS_COMM": "rccl"}), ("buckets", {"DRS_COMM": "rccl", "DRS_RCCL_BUCKETS": "2"})]
    out = str(tmp_path / "double3.npz")
    mp.spawn(_double_worker, args=(3, 29870 + os.getpid() % 1000, out, lib, cases), nprocs=3, join=True)
    r = np.load(out)

    def rel(a, b):
        return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))
    for name in ("inline", "buckets"):
        np.testing.assert_allclose(r[name + "_loss"], r["callback_loss"], rtol=1e-5)
        assert rel(r[name + "_params"], r["callback_params"]) < 1e-4
        assert rel(r[name + "_bn"], r["callback_bn"]) < 1e-5
        assert np.abs(r[name + "_conf"].astype(np.int64) - r["callback_conf"].astype(np.int64)).sum() <= 8
    for what in ("params", "bn", "conf", "loss"):      # (the two library-side forms add the same operands in the same order)
        np.testing.assert_array_equal(r["buckets_" + what], r["inline_" + what])


# ---------------------------------------------------------------------------------------------------------------------
# data parallelism behind the reference's command line (isprs:1987-2138): two processes through cli.main, placed by the
# launcher's environment (dist.from_env), against the single-process run.
ARGV = ["isprs_dilated_random.py", "synthetic:70x80x5/vaihingen/", "OUT", "none", "a,b", "c", "0.01", "0.005", "4", "3", "25", "10",
        "dilated8_grsl", "multi_fixed", "9,13", "acc", "training"]


def _cli_worker(rank, world, port, root):
    import random
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      DRS_DIST_REHEARSAL="1")               # both ranks on the one GPU of the test box, over gloo
    import torch.distributed as dist
    from drs_amd import cli
    os.chdir(root)                                           # the reference keeps its .npy caches in the cwd
    random.seed(40 + rank)                                   # deliberately different: loops.sync_rng must align the ranks
    np.random.seed(50 + rank)
    argv = list(ARGV)
    argv[2] = os.path.join(root, "dp_")
    net = cli.main(argv)                                     # device / communicator from the environment
    assert net.comm.world == 2 and net.b_max == 2
    torch.cuda.synchronize()
    if rank == 0:
        np.save(os.path.join(root, "dp_params.npy"), net.params.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_through_the_command_line_equal_one_process(tmp_path):
    import random
    from drs_amd import cli
    from drs_amd.net import NoComm

    class _SyncOnly(NoComm):
        """single-process stand-in that re-seeds at the loops' synchronisation points exactly like a 2-rank run's rank 0"""
        sync_rng = True
    root2, root1 = str(tmp_path / "two"), str(tmp_path / "one")
    os.makedirs(root2)
    os.makedirs(root1)
    env0 = dict(os.environ)
    try:
        mp.spawn(_cli_worker, args=(2, 30100 + os.getpid() % 1000, root2), nprocs=2, join=True)
    finally:
        os.environ.clear()
        os.environ.update(env0)
    cwd = os.getcwd()
    os.chdir(root1)
    try:
        random.seed(40)
        np.random.seed(50)
        argv = list(ARGV)
        argv[2] = os.path.join(root1, "sp_")
        net = cli.main(argv, device="cuda:0", comm=_SyncOnly())
    finally:
        os.chdir(cwd)
    torch.cuda.synchronize()
    # only rank 0 wrote: one checkpoint, one set of score files, one set of cwd caches, no temporary left behind
    files = sorted(os.listdir(root2))
    assert "dp_model-3.npz" in files and "dp_patch_occur_step_3.npy" in files and not [f for f in files if ".tmp" in f]
    np.testing.assert_array_equal(np.load(os.path.join(root2, "dp_patch_occur_step_3.npy")), np.load(os.path.join(root1, "sp_patch_occur_step_3.npy")))
    np.testing.assert_allclose(np.load(os.path.join(root2, "dp_patch_acc_loss_step_3.npy")),
                               np.load(os.path.join(root1, "sp_patch_acc_loss_step_3.npy")), rtol=0, atol=0.02)
    np.testing.assert_array_equal(np.load(os.path.join(root2, "dataset_vaihingen.npy")), np.load(os.path.join(root1, "dataset_vaihingen.npy")))
    a, b = np.load(os.path.join(root2, "dp_params.npy")), net.params.cpu().numpy()
    assert float(np.abs(a - b).max() / np.abs(b).max()) < 5e-4          # same patches, sizes and augmentation; sums grouped differently


# the coffee / contest command lines under the launcher (coffee:1106-1150, contest:1229-1271): flip-by-index sampling, float16
# patches, void mask -- two ranks against one process
FLAVOUR_ARGV = {
    "coffee": ["coffee_dilated_random.py", "synthetic:2x60x60x3/", "synthetic:1x60x60x3/", "OUT", "none", "0.01", "0.001", "6", "3", "25", "10",
               "dilated_icpr_rate6_small", "multi_fixed", "9,13", "loss"],
    "contest": ["contest_dilated_random.py", "synthetic:80x70x3/", "OUT", "none", "0.01", "0.001", "4", "3", "25", "10", "dilated_grsl",
                "multi_fixed", "9,13", "acc", "train"],
}


def _flavour_main(which):
    from drs_amd import cli
    return cli.main_coffee if which == "coffee" else cli.main_contest


def _flavour_worker(rank, world, port, root, which):
    import random
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      DRS_DIST_REHEARSAL="1")
    import torch.distributed as dist
    os.chdir(root)
    random.seed(60 + rank)
    np.random.seed(70 + rank)
    argv = list(FLAVOUR_ARGV[which])
    argv[argv.index("OUT")] = os.path.join(root, "dp_")
    net = _flavour_main(which)(argv)
    assert net.comm.world == 2
    torch.cuda.synchronize()
    if rank == 0:
        np.save(os.path.join(root, "dp_params.npy"), net.params.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("which", ["coffee", "contest"])
def test_two_ranks_through_the_coffee_and_contest_command_lines(tmp_path, which):
    import random
    from drs_amd.net import NoComm

    class _SyncOnly(NoComm):
        sync_rng = True
    root2, root1 = str(tmp_path / "two"), str(tmp_path / "one")
    os.makedirs(root2)
    os.makedirs(root1)
    env0 = dict(os.environ)
    try:
        mp.spawn(_flavour_worker, args=(2, 30300 + os.getpid() % 1000, root2, which), nprocs=2, join=True)
    finally:
        os.environ.clear()
        os.environ.update(env0)
    cwd = os.getcwd()
    os.chdir(root1)
    try:
        random.seed(60)
        np.random.seed(70)
        argv = list(FLAVOUR_ARGV[which])
        argv[argv.index("OUT")] = os.path.join(root1, "sp_")
        net = _flavour_main(which)(argv, device="cuda:0", comm=_SyncOnly())
    finally:
        os.chdir(cwd)
    torch.cuda.synchronize()
    files = sorted(os.listdir(root2))
    assert "dp_model-3.npz" in files and not [f for f in files if ".tmp" in f]
    a, b = np.load(os.path.join(root2, "dp_params.npy")), net.params.cpu().numpy()
    assert float(np.abs(a - b).max() / np.abs(b).max()) < 1e-3          # same patches, sizes and augmentation; sums grouped differently


# ---------------------------------------------------------------------------------------------------------------------
# band-partitioned sliding-window inference: window rows cut into one band per rank, boundary rows exchanged,
# uint8 label bands gathered -- against the single-process map.
def _tile_for_bands():
    from drs_amd.synthetic import make_tile
    return make_tile(150, 97, 5, 6, seed=12, n_seeds=30)


def _band_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from drs_amd import loops, patches as P
    from drs_amd.dist import TorchComm
    from drs_amd.net import DilatedNet
    torch.cuda.set_device(0)
    comm = TorchComm("gloo")
    tile, lab = _tile_for_bands()
    d = DilatedNet("dilated_grsl", 5, 6, 0.005, b_max=5, s_max=25, device="cuda:0", seed=4, comm=comm)
    pool = P.TilePool([tile], [lab], "cuda:0")
    pred, total = loops.predict_tile(d, pool, 0, 25, 5, [0.5] * 3, [0.2] * 3, comm)
    torch.cuda.synchronize()
    if rank == 1:                      # any rank holds the whole map
        np.save(out, pred.cpu().numpy())
    comm.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 5])       # (5 ranks + this process: as many as a one-GPU box lets share its card; an 8-GPU node runs 8 bands)
def test_band_partitioned_inference_equals_single_process(tmp_path, world):
    from drs_amd import loops, patches as P
    from drs_amd.net import DilatedNet
    out = str(tmp_path / "bands.npy")
    mp.spawn(_band_worker, args=(world, 30600 + os.getpid() % 1000 + world, out), nprocs=world, join=True)
    tile, lab = _tile_for_bands()
    d = DilatedNet("dilated_grsl", 5, 6, 0.005, b_max=5, s_max=25, device="cuda:0", seed=4)
    pool = P.TilePool([tile], [lab], "cuda:0")
    prob, occ, total = loops.predict_tile(d, pool, 0, 25, 5, [0.5] * 3, [0.2] * 3, return_sums=True)
    pred, _ = loops.predict_tile(d, pool, 0, 25, 5, [0.5] * 3, [0.2] * 3)
    torch.cuda.synchronize()
    got = np.load(out)
    want = pred.cpu().numpy()
    avg = (prob.view(150, 97, 6) / occ.view(150, 97, 1).float()).cpu().numpy()
    srt = np.sort(avg, axis=2)
    clear = (srt[..., -1] - srt[..., -2]) > 1e-5 * np.abs(avg).max()
    assert clear.mean() > 0.99 and got.shape == want.shape
    np.testing.assert_array_equal(got[clear], want[clear])        # the float sums associate differently at band boundaries only
    assert (got != want).mean() < 1e-3


# ---------------------------------------------------------------------------------------------------------------------
# rehearsals at the rank counts of the BASELINE configurations, as far as one GPU allows (a GPU box admits 6 processes on its card,
# the test runner included): configs[2]'s command line with the per-rank batch of the 8-GPU run (16 patches, `uniform` over [25, 85]) on 5 ranks, and
# configs[3] (DenseDilated6, `multinomial`, update_type=loss, coffee tiles) on its own 4 ranks.  All ranks on cuda:0 over gloo.
CONFIG3_ARGV = ["isprs_dilated_random.py", "synthetic:200x220x5/vaihingen/", "OUT", "none", "a", "c", "0.01", "0.005", "80", "3", "25", "10",
                "dilated_grsl_rate8", "uniform", "25,85", "acc", "training"]
CONFIG4_ARGV = ["coffee_dilated_random.py", "synthetic:2x120x120x4/", "synthetic:1x120x120x4/", "OUT", "none", "0.01", "0.001", "16", "3", "25", "10",
                "dilated_icpr_rate6_densely", "multinomial", "25,50,75,100", "loss"]


def _rehearsal_worker(rank, world, port, root, which):
    import random
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      DRS_DIST_REHEARSAL="1")
    import torch.distributed as dist
    from drs_amd import cli
    os.chdir(root)
    random.seed(80 + rank)
    np.random.seed(90 + rank)
    argv = list(CONFIG3_ARGV if which == "config3" else CONFIG4_ARGV)
    argv[argv.index("OUT")] = os.path.join(root, "dp_")
    net = (cli.main if which == "config3" else cli.main_coffee)(argv)
    assert net.comm.world == world and net.b_max == int(argv[8 if which == "config3" else 7]) // world
    torch.cuda.synchronize()
    np.save(os.path.join(root, "params_r%d.npy" % rank), net.params.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("which,world", [("config3", 5), ("config4", 4)])
def test_rank_counts_of_the_baseline_configurations_rehearsed_on_one_gpu(tmp_path, which, world):
    root = str(tmp_path)
    env0 = dict(os.environ)
    try:
        mp.spawn(_rehearsal_worker, args=(world, 30900 + os.getpid() % 1000, root, which), nprocs=world, join=True)
    finally:
        os.environ.clear()
        os.environ.update(env0)
    files = sorted(os.listdir(root))
    assert "dp_model-3.npz" in files and not [f for f in files if ".tmp" in f]
    side = "dp_patch_occur_step_3.npy" if which == "config3" else "dp_errorOccur_step_3.npy"
    occ = np.load(os.path.join(root, side))
    assert occ.sum() == 3 and len(occ) == (61 if which == "config3" else 76)          # every step scored its size once, on rank 0 only
    ref = np.load(os.path.join(root, "params_r0.npy"))
    assert np.isfinite(ref).all()
    for r in range(1, world):                                                           # replicated optimizer: every rank holds the same variables
        np.testing.assert_array_equal(np.load(os.path.join(root, "params_r%d.npy" % r)), ref)
