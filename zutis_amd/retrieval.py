"""Retrieval step of the index-dataset pipeline.

Reference (datasets/index_dataset.py:158-167): `text[C,E] @ image[N,E].T` in fp32, then for every category a FULL argsort
of the N similarities (N ~ 2.7 M) of which the first n_images=500 are kept.  Here: the similarity GEMM at the reference's
precision (zh_gemm_f16x3 on split-pair operands: fp32-class scores, so the selected indices are those of the fp32 product
wherever two scores differ by more than fp32 summation-order noise) in column chunks that fit HBM comfortably, and an exact
radix top-k per row (zh_topk_rows) whose winners land in a [C, chunks*k] candidate table; one more top-k over that table
(its idx_map form) merges them.  Ties are broken by ascending image index.  Every step is a libzutis_hip kernel.
"""
from __future__ import annotations

from typing import Tuple

import torch

from . import ops
from ._lib import ZutisHipError
from .ops import Act

f16, f32 = torch.float16, torch.float32


def _split_rows(x: torch.Tensor) -> Act:
    """fp32 [rows, E] on the GPU -> split-pair Act (one cast kernel)."""
    rows, E = x.shape
    a = Act.empty((rows, E), True, x.device)
    # the embeddings are unit-norm rows (extract_image_embeddings.py:73, extract_text_embeddings.py:110-112: elements ~0.04): stored times
    # 2^6 the lo halves of the pairs are normal fp16 numbers (the similarity GEMM multiplies 2^-12 back in); |x| < 1000 cannot overflow
    a.out_scale = 1.0 / 64.0
    ops.cast_f16(x.detach().to(f32).contiguous(), a, rows, E)
    return a


@torch.no_grad()
def retrieve_topk(text_embeddings: torch.Tensor, image_embeddings: torch.Tensor, k: int = 500, chunk: int = 1 << 20,
                  index_offset: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """text [C,E] f32, image [N,E] f32 (both on the GPU, E % 64 == 0) -> (indices int64 [C,k'], scores f32 [C,k']),
    k' = min(k, N); indices are offset by index_offset (a rank's shard of a larger index)."""
    C, E = text_embeddings.shape
    N = image_embeddings.shape[0]
    if E % 64 or image_embeddings.shape[1] != E:
        raise ZutisHipError("retrieve_topk: embedding width must match and be a multiple of 64")
    k = min(k, N)
    chunk = max(8, chunk // 8 * 8)
    t = _split_rows(text_embeddings)
    nchunks = (N + chunk - 1) // chunk
    kc = [min(k, min(chunk, N - j * chunk)) for j in range(nchunks)]
    dev = text_embeddings.device
    if nchunks == 1:
        cand_idx = torch.empty((C, k), dtype=torch.int64, device=dev)
        cand_val = torch.empty((C, k), dtype=f32, device=dev)
    else:                                   # short final chunks leave -inf padding columns that never win
        cand_idx = torch.full((C, sum(kc)), -1, dtype=torch.int64, device=dev)
        cand_val = torch.full((C, sum(kc)), float("-inf"), dtype=f32, device=dev)
    col = 0
    for j in range(nchunks):
        lo = j * chunk
        n = min(chunk, N - lo)
        img = _split_rows(image_embeddings[lo:lo + n])
        scores = torch.empty((C, (n + 7) // 8 * 8), dtype=f32, device=dev)
        ops.gemm_x3(t, img, scores, N=n, fixed_k_order=True)           # text @ image.T   (index_dataset.py:163); shards of any size: same bits
        ops.topk_rows(scores, kc[j], N=n, with_values=True, idx_add=lo + index_offset,
                      out_idx=cand_idx[:, col:col + kc[j]], out_val=cand_val[:, col:col + kc[j]])
        col += kc[j]
    if nchunks == 1:
        return cand_idx, cand_val
    return merge_topk(cand_idx, cand_val, k)        # chunk-major candidates: column ties == image-index ties


def merge_topk(cand_idx: torch.Tensor, cand_val: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Exact top-k over candidate lists: cand_idx int64 [C,n] (global image indices, -1 = padding with score -inf),
    cand_val f32 [C,n] -> (indices [C,k], scores [C,k]), score descending, ties by ascending column.  The lists must be
    concatenations, in ascending index-range order (chunk-major / rank-major), of lists sorted by (score desc, index asc):
    then equal scores appear in ascending image-index order and the result is what retrieve_topk gives over the union."""
    return ops.topk_rows(cand_val.contiguous(), k, with_values=True, idx_map=cand_idx.contiguous())


@torch.no_grad()
def retrieve_topk_sharded(text_embeddings: torch.Tensor, local_image_embeddings: torch.Tensor, index_offset: int, k: int = 500,
                          group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Config-5 retrieval over rank-sharded image embeddings: every rank takes the exact top-k of ITS shard
    (global index = index_offset + local index; shards must be contiguous and ascending with rank), the [C,k] (index, score)
    candidates are all-gathered — 12 B x C x k per rank, 5.5 MB at 919 categories x 500 — and merged identically on every
    rank.  Equals retrieve_topk over the concatenation, including k clamped to the GLOBAL image count."""
    import torch.distributed as dist
    C = text_embeddings.shape[0]
    n_local = local_image_embeddings.shape[0]
    dev = text_embeddings.device
    n_all = torch.tensor([n_local], dtype=torch.int64, device=dev)
    dist.all_reduce(n_all, group=group)
    k = min(k, int(n_all.item()))
    idx = torch.full((C, k), -1, dtype=torch.int64, device=dev)
    val = torch.full((C, k), float("-inf"), dtype=f32, device=dev)
    if n_local > 0:
        li, lv = retrieve_topk(text_embeddings, local_image_embeddings, min(k, n_local), index_offset=index_offset)
        idx[:, : li.shape[1]] = li
        val[:, : lv.shape[1]] = lv
    world = dist.get_world_size(group)
    all_idx = torch.empty((world * C, k), dtype=torch.int64, device=dev)       # rank-major concatenation along dim 0
    all_val = torch.empty((world * C, k), dtype=f32, device=dev)
    dist.all_gather_into_tensor(all_idx, idx, group=group)
    dist.all_gather_into_tensor(all_val, val, group=group)
    all_idx, all_val = all_idx.view(world, C, k), all_val.view(world, C, k)
    return merge_topk(all_idx.permute(1, 0, 2).reshape(C, world * k), all_val.permute(1, 0, 2).reshape(C, world * k), k)
