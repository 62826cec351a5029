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
