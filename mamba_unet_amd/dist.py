"""One-process-per-GPU helpers (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU).

The hot path shards over the batch of 2-D slices with no data-path collective; the only
exchange in training is the gradient all-reduce of DistributedDataParallel.  The reference itself has
no working multi-process path (utils/util.py:236-240 defines an unused helper)."""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> Tuple[int, int, int]:
    """RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* come from the launcher. Returns (rank, world, local_rank)."""
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, world, local


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous shard of `n_items` slices for `rank`; sizes differ by at most one, union is exact."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def max_over_ranks(value: float, device=None) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class _AllReduceSum(torch.autograd.Function):
    """y = sum_r x_r on every rank.  Every rank then evaluates its own copy of the global loss, so the
    adjoint is the sum over ranks of the incoming gradients (world_size * g when the copies agree); after
    DistributedDataParallel's gradient AVERAGE this yields exactly the single-process gradient."""

    @staticmethod
    def forward(ctx, x):
        y = x.clone()
        dist.all_reduce(y, op=dist.ReduceOp.SUM)
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        return g


def all_reduce_sum(x: torch.Tensor) -> torch.Tensor:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return x
    return _AllReduceSum.apply(x)


def wrap_ddp(model: torch.nn.Module, local_rank: int | None = None, force: bool = False):
    """DistributedDataParallel with the settings SURVEY.md 5 derives: every parameter gets a gradient each
    step (find_unused_parameters=False, static_graph=True), buckets are views (one less copy of the 76.5 MB gradient).
    A world of one rank returns the bare model unless `force` (the world-1 RCCL test wraps the real network to run
    DDP's hooks, bucket views and static-graph bookkeeping against this package's autograd nodes on one GPU)."""
    if not (dist.is_available() and dist.is_initialized()):
        if force:
            raise RuntimeError("wrap_ddp(force=True) needs an initialised process group")
        return model
    if dist.get_world_size() == 1 and not force:
        return model
    from torch.nn.parallel import DistributedDataParallel as DDP
    ids = [local_rank] if (local_rank is not None and next(model.parameters()).is_cuda) else None
    # static_graph: the set of used parameters and the order of their gradients never change between iterations
    return DDP(model, device_ids=ids, gradient_as_bucket_view=True, find_unused_parameters=False, static_graph=True)
