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


def payload_checksum(t: torch.Tensor) -> torch.Tensor:
    """Two wrapping int64 sums over the BITS of `t` (4-byte elements): the plain sum and a position-weighted one (a permutation of
    rows, a stale or a zeroed slice all change it).  [2] int64 on t's device."""
    v = t.contiguous().view(torch.int32).to(torch.int64).flatten()
    w = torch.arange(v.numel(), device=v.device, dtype=torch.int64) % 65521 + 1
    return torch.stack([v.sum(), (v * w).sum()])


def verify_gather(lane: "Lane", group=None) -> dict:
    """Did the all-gather of `lane` (retired: call after StepPipeline.drain) really see every rank?  Every rank checksums its own
    payload of the lane's last step; the [world, 2] checksums are all-gathered on their own; slice r of `lane.gathered` must
    checksum to rank r's entry — on EVERY rank (the verdict is min-reduced) — and, when the ranks ran different data, the entries
    differ from one another (a gather that only ever saw this rank's buffer would pass a self-comparison).  New code: the
    reference evaluates on a single device (main.py:54) and has no collective to compare with."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    b = lane.payload.shape[0]
    assert lane.pending is None and lane.gathered is not None and lane.gathered.shape[0] == world * b, "verify_gather: drain the pipeline first"
    mine = payload_checksum(lane.payload)
    flat = torch.empty((world * 2,), dtype=torch.int64, device=mine.device)
    dist.all_gather_into_tensor(flat, mine, group=group)
    sums = flat.view(world, 2)
    got = torch.stack([payload_checksum(lane.gathered[r * b:(r + 1) * b]) for r in range(world)])
    ok = torch.tensor([int(torch.equal(got, sums))], dtype=torch.int64, device=mine.device)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    rows = [tuple(r) for r in sums.tolist()]
    return {"ranks_in_gather": world, "verified": bool(ok.item() == 1), "slices_distinct": len(set(rows)) == world,
            "bytes_per_rank": int(lane.payload.numel() * lane.payload.element_size()), "step": int(lane.step),
            "mismatching_slices_on_this_rank": [r for r in range(world) if not torch.equal(got[r], sums[r])]}


class Lane:
    """One evaluation step in flight: `payload` is the tensor the step's kernels write and the collective reads (low-res class
    logits), `gathered` the all-gather destination, `stream` the HIP stream its work is enqueued on (None on CPU), `state`
    whatever the launcher needs (engine, launch plan, outputs)."""
    __slots__ = ("payload", "gathered", "stream", "state", "pending", "step")

    def __init__(self, payload, gathered=None, stream=None, state=None):
        self.payload, self.gathered, self.stream, self.state = payload, gathered, stream, state
        self.pending, self.step = None, -1


class StepPipeline:
    """Scheduling of an evaluation loop with `len(lanes)` independent steps in flight (bench.py, config 2 / 4 at N GPUs).

    Steps are processed in groups of up to n_lanes: `launch(group, step_ids)` enqueues the compute of the group's lanes
    (interleaved on their streams — zutis_amd.plan.run_many on the GPU), then every lane issues an ASYNC all-gather of its
    payload on its own stream, so the collective of step i overlaps the compute of the steps behind it.  A lane's payload and
    gather buffer are reused every n_lanes steps: before the lane is launched again its previous gather is waited for (on the
    lane's stream: the wait orders the stream, not the host) and `consume(lane, step_id)` — if given — sees the gathered
    result of that earlier step.  The final group may be ragged (count % n_lanes lanes)."""

    def __init__(self, lanes, launch, gather: bool, group=None, consume=None):
        self.lanes, self.launch, self.gather, self.group, self.consume = list(lanes), launch, gather, group, consume
        self.next_step = 0
        self.retired = 0                      # gathers waited for (== steps run once drained)

    def _on_stream(self, lane):
        import contextlib
        return torch.cuda.stream(lane.stream) if lane.stream is not None else contextlib.nullcontext()

    def _retire(self, lane):
        if lane.pending is not None:
            with self._on_stream(lane):
                lane.pending.wait()
            lane.pending = None
            self.retired += 1
            if self.consume is not None:
                self.consume(lane, lane.step)

    def run(self, count: int):
        done = 0
        while done < count:
            grp = self.lanes[:min(len(self.lanes), count - done)]
            ids = list(range(self.next_step, self.next_step + len(grp)))
            if self.gather:
                for ln in grp:                 # the lane's previous gather must have read `payload` before it is rewritten
                    self._retire(ln)
            self.launch(grp, ids)
            for ln, i in zip(grp, ids):
                ln.step = i
                if self.gather:
                    with self._on_stream(ln):
                        ln.gathered, ln.pending = all_gather_logits(ln.payload, out=ln.gathered, async_op=True, group=self.group)
            done += len(grp)
            self.next_step += len(grp)

    def drain(self):
        for ln in self.lanes:
            self._retire(ln)
