"""-m gpu: the collective calls the data-parallel step issues (dist.py / net.train_step), through the RCCL backend itself.
A one-GPU box cannot form a multi-rank RCCL group, so this is a single-rank group: it checks that the backend accepts
exactly the tensor kinds, slices and asynchronous forms the step uses (the arithmetic of the sharded step is covered by
test_dist_gloo.py on CPU and test_gpu_dp.py on the GPU over gloo)."""
import os

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def test_rccl_accepts_the_collectives_of_the_step():
    assert torch.cuda.is_available()
    if dist.is_initialized():
        pytest.skip("a process group is already up in this process")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))     # as dist.TorchComm does
    try:
        dev = "cuda:0"
        sums = torch.arange(512, dtype=torch.float64, device=dev)           # batch-norm statistics: a slice of an fp64 buffer
        h = dist.all_reduce(sums[:2 * 192], op=dist.ReduceOp.SUM, async_op=True)
        grads = torch.ones(2_091_590, dtype=torch.float32, device=dev)      # a gradient bucket: an interior slice of the flat buffer
        hb = dist.all_reduce(grads[1000:900_000], op=dist.ReduceOp.SUM, async_op=True)
        h.wait()
        conf = torch.ones(36, dtype=torch.int32, device=dev)
        dist.all_reduce(conf, op=dist.ReduceOp.SUM)
        scal = torch.ones(4, dtype=torch.float64, device=dev)
        dist.all_reduce(scal[:1], op=dist.ReduceOp.SUM)
        t = torch.tensor([1.5], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        hb.wait()
        dist.barrier()
        torch.cuda.synchronize()
        assert float(sums[383]) == 383.0 and float(grads[5000]) == 1.0 and int(conf[0]) == 1 and float(t) == 1.5
    finally:
        dist.destroy_process_group()
