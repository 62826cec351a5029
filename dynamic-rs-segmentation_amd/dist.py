"""Data parallelism: one process per GPU, torch.distributed over RCCL/xGMI ('nccl' IS RCCL on ROCm).

The reference is single-process (isprs:1707); this is the build's addition.  A training
step shards its batch over the ranks (same patch size, same RNG streams on every rank, so all ranks agree on
the size draw, the batch indices and the augmentation without any exchange) and needs exactly these
collectives, all sums:
  * per batch-norm layer, forward:  [sum z, sum z^2]      (2*C fp64)  -> the reference's global-batch statistics
  * per batch-norm layer, backward: [sum g, sum g*xhat]   (2*C fp64)
  * once per step: the flat fp32 gradient buffer (8.37 MB for Dilated8Pooling), the CE sum (1 fp64) and the
    K x K confusion matrix (int32)
Sliding-window inference shards windows over ranks with no data-path collective until the final band gather.
"""
import os

import torch
import torch.distributed as dist


class TorchComm(object):
    """all_reduce_sum over the default process group (backend 'nccl' on GPUs, 'gloo' on CPU tests)."""

    def __init__(self, backend=None, init=True):
        if init and not dist.is_initialized():
            backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend == "nccl":       # bind the communicator to this rank's GPU at once (the caller has already set the device)
                dist.init_process_group(backend=backend, device_id=torch.device("cuda", torch.cuda.current_device()))
            else:
                dist.init_process_group(backend=backend)
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.backend = dist.get_backend()
        # DRS_FORCE_COLLECTIVES=1: issue every collective of the step at world 1 too (sums over one rank: identities).  A one-GPU box
        # can then drive the real RCCL path -- communicator bound to the device, async work handles, stream waits -- end to end.
        self.collective = self.world > 1 or os.environ.get("DRS_FORCE_COLLECTIVES") == "1"
        # The collectives of one communicator run in order on ITS stream: a 2 KB sync-BN sum issued behind a 4 MB gradient bucket
        # waits for the bucket.  In the step's schedule a bucket (~0.15 ms on 8 GPUs) is followed by >= 0.35 ms of kernels before the
        # next batch-norm sum is issued, so the queue is normally empty by then and ONE communicator is the default.  DRS_BN_COMM=1
        # gives the latency-bound sums (<= SMALL elements) a communicator of their own (they then overtake the buckets); it is opt-in
        # because no multi-GPU node was available to this build to measure either choice.
        two = self.collective and os.environ.get("DRS_BN_COMM") == "1"
        self.small = dist.new_group(ranks=list(range(self.world))) if two else None

    SMALL = 4096

    def _group(self, t):
        return self.small if t.numel() <= self.SMALL else None

    def all_reduce_sum(self, t):
        if self.collective:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self._group(t))
        return t

    def all_reduce_sum_async(self, t):
        """start a sum all-reduce on the collective's own stream (overlaps the kernels enqueued afterwards);
        returns a handle for wait()."""
        if self.collective:
            return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self._group(t), async_op=True)
        return None

    @staticmethod
    def wait(handles):
        """make the current stream wait for the collectives started by all_reduce_sum_async (no host block on nccl)."""
        for h in handles:
            if h is not None:
                h.wait()

    def barrier(self):
        if self.world > 1:
            dist.barrier()

    @property
    def sync_rng(self):
        """the step loops re-seed `random` / `numpy.random` from rank 0 at their synchronisation points (loops.sync_rng)"""
        return self.world > 1

    def broadcast_object(self, obj, src=0):
        """rank `src`'s picklable object on every rank (index tables, cached .npy contents, RNG seeds)"""
        if self.world == 1:
            return obj
        box = [obj if self.rank == src else None]
        dist.broadcast_object_list(box, src=src)
        return box[0]

    def gather_objects(self, obj):
        """every rank's picklable object, in rank order, on every rank (diagnostics: each rank's first-contact timings)"""
        if self.world == 1:
            return [obj]
        box = [None] * self.world
        dist.all_gather_object(box, obj)
        return box

    def agree(self, values, what="value"):
        """raise on every rank unless all ranks hold the same integers (a cheap guard of the lock-step host logic)"""
        if self.world == 1:
            return
        t = torch.tensor([int(v) for v in values], dtype=torch.int64)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        lo, hi = t.clone(), t.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if not torch.equal(lo, hi):
            raise RuntimeError("ranks disagree on %s: min %s max %s (rank %d has %s)" % (what, lo.tolist(), hi.tolist(), self.rank, t.tolist()))

    def all_true(self, flag):
        """True on every rank iff `flag` is true on every rank (collective decisions: which collectives path a net takes)"""
        if self.world == 1:
            return bool(flag)
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    def max_float(self, v, device):
        t = torch.tensor([float(v)], dtype=torch.float64, device=device)
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())


def shard_slice(n, rank, world):
    """contiguous share of n items for `rank` (n must divide evenly: batch-norm counts assume equal shards)."""
    if n % world:
        raise ValueError("global batch %d is not divisible by %d ranks" % (n, world))
    per = n // world
    return slice(rank * per, (rank + 1) * per)


def window_shard(n_windows, batch_size, rank, world):
    """whole batches of consecutive windows per rank, round-robin over batches (inference)."""
    nb = -(-n_windows // batch_size)
    return [i for i in range(nb) if i % world == rank]


def from_env(backend=None):
    """(device, comm) of this process.  Launched by `python -m torch.distributed.run --nproc-per-node N ...` (WORLD_SIZE > 1):
    bind to GPU LOCAL_RANK and form the process group BEFORE any other GPU call, one rank per GPU over RCCL.  Plain launch:
    ("cuda:0", None).  DRS_DIST_REHEARSAL=1 puts every rank on cuda:0 over gloo (one-GPU boxes: exercises the N > 1 code path,
    says nothing about speed)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return "cuda:0", None
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (what RCCL needs here); read when the runtime starts
    rehearsal = os.environ.get("DRS_DIST_REHEARSAL") == "1"
    local = 0 if rehearsal else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    comm = TorchComm(backend or ("gloo" if rehearsal else "nccl"))
    return "cuda:%d" % local, comm
