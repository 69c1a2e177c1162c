"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl") on xGMI.

The reference is single-device (train_lightning.py:46 ``devices=1``; sed.py:42).  The path shards by
minibatch sample: rank r trains on samples [r*B/W, (r+1)*B/W); the only exchange is the
gradient all-reduce, issued per backward stage on the flat arena so it overlaps the conv backward.
BatchNorm uses per-rank batch statistics (what torch DDP does).  Device-agnostic on purpose: the
gloo/CPU tests drive the same class.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun); returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_batch(x, rank, world):
    """Rank r's contiguous share of a global batch (dim 0); the global batch must divide evenly."""
    n = x.shape[0]
    if n % world:
        raise ValueError(f"global batch {n} is not divisible by world size {world}")
    per = n // world
    return x[rank * per:(rank + 1) * per]


class BucketedAllReduce:
    """All-reduce (average) of contiguous slices of a flat gradient arena, one slice per backward stage.

    ``launch(i)`` enqueues slice i asynchronously behind the work already on the current stream (RCCL runs
    it on its own stream, so it overlaps whatever is enqueued next); ``wait_all()`` makes the current
    stream wait for every outstanding slice.  Slices are in backward-completion order.
    """

    def __init__(self, flat_grad, slices, group=None):
        self.flat, self.slices, self.group = flat_grad, list(slices), group
        self.world = dist.get_world_size(group)
        self._work = []
        self._avg = dist.ReduceOp.AVG if flat_grad.is_cuda else None     # gloo has no AVG

    def launch(self, i):
        a, b = self.slices[i]
        if b <= a:
            return
        view = self.flat[a:b]
        if self._avg is not None:
            self._work.append((dist.all_reduce(view, op=self._avg, group=self.group, async_op=True), None))
        else:
            self._work.append((dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True), view))

    def wait_all(self):
        for w, view in self._work:
            w.wait()
            if view is not None:
                view.div_(self.world)
        self._work.clear()


def broadcast_parameters(model, src=0, group=None):
    """Make every rank start from rank ``src``'s weights and BN running statistics."""
    dist.broadcast(model.flat_parameters(), src=src, group=group)
    for b in model.buffers():
        dist.broadcast(b, src=src, group=group)
