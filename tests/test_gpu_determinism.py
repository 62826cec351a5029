"""-m gpu: the step engine is bitwise reproducible from run to run in the regime that runs kernels side by side.

Below 2^18 pixels the backward pass puts the filter gradients on a stream of their own beside the batch-norm backward / input
gradient chain (csrc/engine.hip), so kernels share the chip and their timing varies from run to run; every promise of a fixed
summation order has to survive that.  Two identical runs of a training loop with the patch side drawn per step (isprs:1727-1737:
ragged pixel counts, stream-K cuts, partial last chunks of the filter gradient) must agree bit for bit in every loss and every
variable.  (Round 3: this is the test that found wgrad_dma_kernel zeroing the tail rows of its last chunk before the other waves'
LDS-DMA had landed -- deterministic alone, a run-to-run difference under concurrency.  tools/soak.py is the long form.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import DEV   # noqa: E402


def _run(B, steps, sizes, pool):
    from drs_amd.net import DilatedNet
    from drs_amd import patches as P
    from drs_amd.synthetic import grid_instances
    net = DilatedNet("dilated_grsl_rate8", 5, 6, 0.005, b_max=B, s_max=max(sizes), device=DEV, seed=42)
    rng = np.random.default_rng(7)
    np.random.seed(11)
    inst = {S: grid_instances(512, 512, S, 25, 1024, seed=S) for S in sizes}
    losses = torch.zeros(steps, 2, dtype=torch.float64, device=DEV)
    for i in range(steps):
        S = int(sizes[rng.integers(0, len(sizes))])
        rows = inst[S][(i * B) % 900:(i * B) % 900 + B]
        aug = P.draw_augmentation(rows, S, 5, noise="device")
        P.crop_to_net(net, pool, rows, S, [0.5] * 3, [0.2] * 3, aug)
        losses[i] = net.train_step(B, S, 0.01)["loss_parts"]
    torch.cuda.synchronize()
    state = torch.cat([net.params.flatten(), net.mom.flatten(), net.bn.flatten()]).cpu().numpy()
    return losses.cpu().numpy(), state


@pytest.mark.parametrize("B,sizes,steps", [(16, (25, 33, 38, 45, 55, 61), 160), (8, (75, 85), 60)])
def test_two_runs_of_a_mixed_size_training_loop_agree_bit_for_bit(B, sizes, steps):
    from drs_amd import patches as P
    from drs_amd.synthetic import make_tile
    tile, lab = make_tile(512, 512, 5, 6, seed=1234)
    pool = P.TilePool([tile], [lab], DEV)
    l1, s1 = _run(B, steps, sizes, pool)
    l2, s2 = _run(B, steps, sizes, pool)
    assert np.isfinite(l1).all() and np.isfinite(s1).all()
    diff = np.nonzero((l1 != l2).any(axis=1))[0]
    assert diff.size == 0, "losses differ from step %d on" % diff[0]
    assert np.array_equal(s1, s2)


def _jitter_worker(seed, out):
    """one process on the development library (the same sources + include/drs_dev.h): the mixed-size loop with sleeps of 0 .. 150 us put on
    the step's streams where they hand work to each other (drs_debug_jitter; 0 = none)"""
    from drs_amd import _lib
    from drs_amd import patches as P
    from drs_amd.synthetic import make_tile
    d = _lib.dev()
    _lib._lib = d.lib
    d.drs_debug_jitter(seed)
    tile, lab = make_tile(512, 512, 5, 6, seed=1234)
    pool = P.TilePool([tile], [lab], DEV)
    losses, state = _run(16, 60, (25, 33, 45, 61, 64), pool)
    np.savez(out, losses=losses, state=state)


def test_sleeps_on_the_steps_streams_do_not_change_a_bit(tmp_path):
    """Schedule fuzzing of the two-stream training step (r05: its preparation launch beside the forward pass, the classifier's slab
    reductions, the filter gradients and the chain at wave priorities of their own all hand work across streams): random sleeps at every
    hand-over move each cross-stream dependency off its usual timing; a missing event wait that the usual timing hides would change the
    result.  Two seeds of sleeps against none: every loss and every variable bit for bit."""
    import torch.multiprocessing as mp
    outs = []
    for seed in (0, 12345, 987654321):
        out = str(tmp_path / ("jitter%d.npz" % seed))
        ctx = mp.get_context("spawn")
        pr = ctx.Process(target=_jitter_worker, args=(seed, out))
        pr.start(); pr.join()
        assert pr.exitcode == 0
        outs.append(np.load(out))
    assert np.isfinite(outs[0]["losses"]).all()
    for o in outs[1:]:
        assert np.array_equal(o["losses"], outs[0]["losses"]) and np.array_equal(o["state"], outs[0]["state"])


KEEP = ("params", "momentum", "bn", "labels", "acc_mask", "loss_mask", "w0pad")


def _steps(net_type, channels, classes, B, S, poison, pool):
    from drs_amd.net import DilatedNet
    from drs_amd import patches as P
    from drs_amd.synthetic import grid_instances
    net = DilatedNet(net_type, channels, classes, 0.005, b_max=B, s_max=S, device=DEV, seed=42)
    inst = grid_instances(512, 512, S, 25, 512, seed=3)
    np.random.seed(5)
    outs = []
    for i in range(3):
        rows = inst[i * B:(i + 1) * B]
        aug = P.draw_augmentation(rows, S, channels, noise="device")
        if poison:
            for name, t in net._bufs.items():
                if name in KEEP or name.startswith("act:"):        # state that persists; the activation slabs' zero halos
                    continue
                if t.dtype in (torch.float32, torch.float64):
                    t.fill_(float("nan"))
                else:
                    t.fill_(-1 if t.dtype in (torch.int32, torch.int64) else 255)
        P.crop_to_net(net, pool, rows, S, [0.5] * 3, [0.2] * 3, aug)
        o = net.train_step(B, S, 0.01)
        torch.cuda.synchronize()
        outs.append((o["loss_parts"].clone().cpu().numpy(), o["pred"].clone().cpu().numpy(), o["conf"].clone().cpu().numpy()))
    return outs, torch.cat([net.params.flatten(), net.mom.flatten(), net.bn.flatten()]).cpu().numpy()


@pytest.mark.parametrize("net_type,channels,classes,B,S", [("dilated_grsl_rate8", 5, 6, 16, 37), ("dilated_icpr_rate6_densely", 4, 2, 16, 41),
                                                           ("dilated_icpr_rate6_SE", 3, 6, 8, 30), ("dilated_icpr_rate6_squeeze", 3, 6, 8, 33)])
def test_a_step_reads_no_scratch_it_has_not_written(net_type, channels, classes, B, S):
    """every scratch buffer of the step engine (partial-sum slabs, the stream-K workspace, gradient slabs, raw conv outputs, pool
    positions, gradient / logits / prediction buffers, ...) filled with NaN / 0xFF before each of three steps: same bits as without"""
    from drs_amd import patches as P
    from drs_amd.synthetic import make_tile
    tile, lab = make_tile(512, 512, channels, classes, seed=1234)
    pool = P.TilePool([tile], [lab], DEV)
    a, sa = _steps(net_type, channels, classes, B, S, False, pool)
    b, sb = _steps(net_type, channels, classes, B, S, True, pool)
    assert np.isfinite(sa).all()
    for i, (x, y) in enumerate(zip(a, b)):
        for j in range(3):
            assert np.array_equal(x[j], y[j]), (i, j)
    assert np.array_equal(sa, sb)
