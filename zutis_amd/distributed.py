"""Data-parallel evaluation helpers (new: the reference is single-GPU).

One process per GPU; backend "nccl" is RCCL on ROCm (xGMI), "gloo" in the CPU tests.  Images are independent, so the
only exchange is at the evaluation boundary: an all-gather of the low-resolution class logits (what
predict(return_logits=True) would upsample) or an all-reduce of the n x n confusion matrix.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of rank `rank`: ranks < n % world get one extra item; a gather in rank order
    reproduces the reference's image order."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_logits(local: torch.Tensor, out: Optional[torch.Tensor] = None, async_op: bool = False, group=None):
    """local [b, n, h, w] (equal b on every rank) -> [world*b, n, h, w] in rank order."""
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    work = dist.all_gather_into_tensor(out, local.contiguous(), group=group, async_op=async_op)
    return (out, work) if async_op else out


def all_gather_ragged(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """Shards of unequal length along dim 0 (shard_range split): pad to the max shard, gather, trim."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return torch.cat([out[r * mx: r * mx + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], dim=0)


def all_reduce_confusion(hist: torch.Tensor, group=None) -> torch.Tensor:
    """Sum the per-rank int64 n x n confusion matrices (utils/running_score.py:11-20 accumulates them serially)."""
    dist.all_reduce(hist, op=dist.ReduceOp.SUM, group=group)
    return hist
