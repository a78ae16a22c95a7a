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
