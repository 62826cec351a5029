"""CPU, world_size 2 over gloo: the data-parallel algorithm the HIP path uses (batch shards, sync batch norm through
all-reduced sums in both directions, one flat gradient all-reduce, 1/N_global loss scaling) reproduces the
single-process result.  The arithmetic here is the oracle's (test infrastructure); the collective wrapper, the
sharding helpers and the exchange sequence are the product's (drs_amd.dist; sequence as in drs_amd.net.train_step)."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import tf_ops as T


def _sharded_loss_and_grads(o, x, y, comm, n_global, wd):
    """OracleNet.loss_and_grads with every batch statistic exchanged as all-reduced sums (what each rank does)."""
    def allsum(a):
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64))
        comm.all_reduce_sum(t)
        return t.numpy()
    spec = o.spec
    cur, cache = x, []
    for li, (name, k, ci, co, r) in enumerate(o.convs):
        z = T.conv2d_same(cur, o.p[name + "/weights"], r) + o.p[name + "/biases"]
        s = allsum(np.stack([z.sum(axis=(0, 1, 2)), (z ** 2).sum(axis=(0, 1, 2))]))
        mean = s[0] / n_global
        var = s[1] / n_global - mean ** 2
        rstd = 1 / np.sqrt(var + T.BN_EPS)
        xh = (z - mean) * rstd
        a = T.act_fwd(xh, spec["act"])
        out, idx = T.max_pool_3x3(a) if spec["pool"] else (a, None)
        cache.append((cur, z, rstd, xh, idx))
        cur = out
    logits = cur @ o.p["conv_classifier/weights"][0, 0] + o.p["conv_classifier/biases"]
    K = logits.shape[-1]
    ce_mean_local, gl = T.softmax_ce(logits, y)
    n_local = logits.size // K
    gl = gl * (n_local / n_global)                      # dL/dlogits with the GLOBAL 1/N
    ce_sum = allsum(np.array([ce_mean_local * n_local]))[0] / n_global
    g = {"conv_classifier/weights": (cur.reshape(-1, cur.shape[-1]).T @ gl.reshape(-1, K)).reshape(1, 1, -1, K),
         "conv_classifier/biases": gl.reshape(-1, K).sum(axis=0)}
    gcur = gl @ o.p["conv_classifier/weights"][0, 0].T
    for li in reversed(range(len(o.convs))):
        name, k, ci, co, r = o.convs[li]
        inp, z, rstd, xh, idx = cache[li]
        ga = T.max_pool_3x3_bwd(idx, gcur) if spec["pool"] else gcur
        gxh = T.act_bwd(xh, spec["act"], ga)
        s = allsum(np.stack([gxh.sum(axis=(0, 1, 2)), (gxh * xh).sum(axis=(0, 1, 2))]))
        gz = rstd * (gxh - s[0] / n_global - xh * s[1] / n_global)
        gcur, gw = T.conv2d_same_bwd(inp, o.p[name + "/weights"], r, gz)
        g[name + "/weights"] = gw
    names = sorted(g)
    flat = allsum(np.concatenate([g[n].reshape(-1) for n in names]))        # ONE flat gradient all-reduce
    off = 0
    for n in names:
        g[n] = flat[off:off + g[n].size].reshape(g[n].shape)
        off += g[n].size
        if n.endswith("/weights"):
            g[n] = g[n] + wd * o.p[n]
    return ce_sum, g


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"], os.environ["WORLD_SIZE"] = str(rank), str(world)
    from drs_amd.dist import TorchComm, shard_slice, window_shard
    comm = TorchComm("gloo")
    assert comm.world == world and comm.rank == rank
    net, ch, K, B, S = "dilated_grsl", 3, 6, 4, 9
    rng = np.random.default_rng(0)                      # same streams on every rank
    x = rng.normal(size=(B, S, S, ch))
    y = rng.integers(0, K, size=(B, S, S))
    o = T.OracleNet(net, ch, K, seed=1)
    sl = shard_slice(B, rank, world)
    loss, g = _sharded_loss_and_grads(o, x[sl], y[sl], comm, float(B * S * S), 0.005)
    # every window batch is owned by exactly one rank
    owned = torch.zeros(7, dtype=torch.int32)
    for i in window_shard(50, 8, rank, world):
        owned[i] += 1
    comm.all_reduce_sum(owned)
    assert owned.tolist() == [1] * 7
    if rank == 0:
        np.savez(tmp, loss=loss, **{k.replace("/", "__"): v for k, v in g.items()})
    comm.barrier()
    dist.destroy_process_group()


def test_two_rank_data_parallel_step_equals_single_process(tmp_path):
    out = str(tmp_path / "dp.npz")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    net, ch, K, B, S = "dilated_grsl", 3, 6, 4, 9
    rng = np.random.default_rng(0)
    x = rng.normal(size=(B, S, S, ch))
    y = rng.integers(0, K, size=(B, S, S))
    o = T.OracleNet(net, ch, K, seed=1)
    loss_ref, _, g_ref, _ = o.loss_and_grads(x, y, 0.005)
    l2 = sum(0.5 * (o.p[n] ** 2).sum() for n in o.p if n.endswith("/weights"))
    d = np.load(out)
    assert abs(float(d["loss"]) + 0.005 * l2 - loss_ref) < 1e-10
    for n in g_ref:
        if n.endswith("/weights") or n == "conv_classifier/biases":
            np.testing.assert_allclose(d[n.replace("/", "__")], g_ref[n], rtol=1e-7, atol=1e-10, err_msg=n)


def test_shard_slice_rejects_uneven_batches():
    from drs_amd.dist import shard_slice
    assert shard_slice(128, 3, 8) == slice(48, 64)
    try:
        shard_slice(10, 0, 4)
        assert False
    except ValueError:
        pass


def _sync_worker(rank, world, port, tmp):
    """loops.sync_rng / rank0_cached / TorchComm.agree with two ranks that start from DIFFERENT RNG states and of which only
    rank 0 builds the cache (so only rank 0 consumes draws there): what the data-parallel step loops rely on."""
    import random
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"], os.environ["WORLD_SIZE"] = str(rank), str(world)
    from drs_amd.dist import TorchComm
    from drs_amd import loops
    comm = TorchComm("gloo")
    random.seed(100 + rank)
    np.random.seed(200 + rank)
    loops.sync_rng(comm)
    a = (random.random(), float(np.random.rand()))
    path = os.path.join(tmp, "cache.npy")

    def make():
        assert comm.rank == 0                       # only rank 0 ever builds
        return np.random.randint(0, 1000, size=(5, 3))
    v1 = loops.rank0_cached(comm, path, make)       # built by rank 0 (draws there only), broadcast
    loops.sync_rng(comm)                            # ... so the streams are re-aligned afterwards
    b = (random.random(), float(np.random.rand()))
    v2 = loops.rank0_cached(comm, path, make)       # now loaded by rank 0, broadcast
    comm.agree((int(v1.sum()), int(v2.sum()), 7), "cache contents")
    ok = False
    try:
        comm.agree((rank,), "rank (must differ)")
    except RuntimeError:
        ok = True
    assert ok
    # a failure on rank 0 (a corrupt cache file, a builder that raises) fails EVERY rank instead of stranding the others in the broadcast
    bad = os.path.join(tmp, "corrupt.npy")
    if rank == 0:
        open(bad, "wb").write(b"not an npy file")
    comm.barrier()
    for trial in (lambda: loops.rank0_cached(comm, bad, make),
                  lambda: loops.rank0_cached(comm, os.path.join(tmp, "never.npy"), lambda: 1 / 0)):
        raised = False
        try:
            trial()
        except RuntimeError as e:
            raised = "rank 0 failed" in str(e)
        assert raised
    np.save(os.path.join(tmp, "r%d.npy" % rank), np.array([a[0], a[1], b[0], b[1], float(v1.sum()), float(v2.sum())]))
    comm.barrier()
    dist.destroy_process_group()


def test_rng_and_cache_synchronisation_across_ranks(tmp_path):
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_sync_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "r0.npy"), np.load(tmp_path / "r1.npy")
    np.testing.assert_array_equal(r0, r1)
    assert os.path.isfile(tmp_path / "cache.npy") and not [f for f in os.listdir(tmp_path) if ".tmp" in f]


def test_band_plan_covers_every_row_once():
    """loops.band_plan: bands of consecutive ranks overlap (windows overlap by S - stride), owned row ranges partition [0, h)."""
    from drs_amd import patches as P
    from drs_amd.loops import band_plan
    for h, S, W in [(6000, 64, 8), (200, 25, 3), (130, 64, 2), (70, 25, 4), (97, 25, 2), (1000, 85, 8)]:
        st = S // 2
        n_h, _ = P.window_counts(h, h, S, st)
        a, top, bot, own = band_plan(h, S, st, n_h, W)
        assert a[0] == 0 and a[-1] == n_h and all(a[i] < a[i + 1] for i in range(W))
        assert own[0] == 0 and own[-1] == h and all(own[i] < own[i + 1] for i in range(W))
        for r in range(W):
            assert top[r] <= own[r] and bot[r] >= own[r + 1] and bot[r] <= h          # a rank's band holds every row it owns
            rows = {min(i * st, h - S) for i in range(a[r], a[r + 1])}
            assert min(rows) == top[r] and max(rows) + S == bot[r]
