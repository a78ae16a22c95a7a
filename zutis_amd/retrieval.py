"""Retrieval step of the index-dataset pipeline.

Reference (datasets/index_dataset.py:158-167): `text[C,E] @ image[N,E].T`, then for every category a FULL argsort of the
N similarities (N ~ 2.7 M) of which the first n_images=500 are kept.  Here: the similarity GEMM on fp16 MFMA (fp32
accumulate) in column chunks that fit HBM comfortably, and an exact radix top-k per row (zh_topk_rows); chunk winners are
merged by one more top-k over the C x (chunks*k) candidates.  Ties are broken by ascending image index.
"""
from __future__ import annotations

from typing import Tuple

import torch

from . import ops

f16, f32 = torch.float16, torch.float32


@torch.no_grad()
def retrieve_topk(text_embeddings: torch.Tensor, image_embeddings: torch.Tensor, k: int = 500,
                  chunk: int = 1 << 20) -> Tuple[torch.Tensor, torch.Tensor]:
    """text [C,E] f32, image [N,E] f32/f16 (both on the GPU, E % 64 == 0) -> (indices int64 [C,k], scores f32 [C,k])."""
    C, E = text_embeddings.shape
    N = image_embeddings.shape[0]
    k = min(k, N)
    t16 = text_embeddings.to(f16).contiguous()
    cand_idx, cand_val = [], []
    for lo in range(0, N, chunk):
        n = min(chunk, N - lo)
        npad = (n + 7) // 8 * 8
        img = torch.zeros((npad, E), dtype=f16, device=t16.device)
        img[:n] = image_embeddings[lo:lo + n].to(f16)
        scores = torch.empty((C, npad), dtype=f32, device=t16.device)
        ops.gemm(t16, img, scores)                                    # text @ image.T   (index_dataset.py:163)
        kk = min(k, n)
        idx, val = ops.topk_rows(scores, kk, N=n, with_values=True)
        cand_idx.append(idx + lo)
        cand_val.append(val)
    if len(cand_idx) == 1:
        return cand_idx[0], cand_val[0]
    ci, cv = torch.cat(cand_idx, 1), torch.cat(cand_val, 1).contiguous()   # candidates are in ascending-index chunk order,
    sel, val = ops.topk_rows(cv, k, with_values=True)                       # so position ties == index ties
    return torch.gather(ci, 1, sel), val


def merge_topk(cand_idx: torch.Tensor, cand_val: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Exact top-k over per-shard candidates: cand_idx int64 [C,n] (global image indices, -1 = padding), cand_val f32 [C,n]
    -> (indices [C,k], scores [C,k]), score descending, ties by ascending image index — the order retrieve_topk produces.
    Host-side bookkeeping on C x (world*k) numbers (plain torch ops, any device)."""
    val = torch.where(cand_idx < 0, torch.full_like(cand_val, float("-inf")), cand_val)
    o1 = torch.argsort(cand_idx, dim=1, stable=True)                       # ascending index first ...
    ci, cv = torch.gather(cand_idx, 1, o1), torch.gather(val, 1, o1)
    o2 = torch.argsort(cv, dim=1, descending=True, stable=True)[:, :k]     # ... so the stable score sort keeps index order on ties
    return torch.gather(ci, 1, o2), torch.gather(cv, 1, o2)


@torch.no_grad()
def retrieve_topk_sharded(text_embeddings: torch.Tensor, local_image_embeddings: torch.Tensor, index_offset: int, k: int = 500,
                          group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Config-5 retrieval over rank-sharded image embeddings: every rank takes the exact top-k of ITS shard
    (global index = index_offset + local index), the [C,k] (index, score) candidates are all-gathered — 8 B x C x k per rank,
    3.7 MB at 919 categories x 500 — and merged identically on every rank.  Equals retrieve_topk over the concatenation."""
    import torch.distributed as dist
    C = text_embeddings.shape[0]
    n_local = local_image_embeddings.shape[0]
    dev = text_embeddings.device
    idx = torch.full((C, k), -1, dtype=torch.int64, device=dev)
    val = torch.full((C, k), float("-inf"), dtype=f32, device=dev)
    if n_local > 0:
        li, lv = retrieve_topk(text_embeddings, local_image_embeddings, min(k, n_local))
        idx[:, : li.shape[1]] = li + index_offset
        val[:, : lv.shape[1]] = lv
    world = dist.get_world_size(group)
    all_idx = torch.empty((world * C, k), dtype=torch.int64, device=dev)       # rank-major concatenation along dim 0
    all_val = torch.empty((world * C, k), dtype=f32, device=dev)
    dist.all_gather_into_tensor(all_idx, idx, group=group)
    dist.all_gather_into_tensor(all_val, val, group=group)
    all_idx, all_val = all_idx.view(world, C, k), all_val.view(world, C, k)
    return merge_topk(all_idx.permute(1, 0, 2).reshape(C, world * k), all_val.permute(1, 0, 2).reshape(C, world * k), k)
