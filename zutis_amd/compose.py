"""Pack-time algebra of the ZUTIS head: exact rewrites of reference expressions, evaluated in fp64 and rounded
to fp32 once.  Pure tensor functions (any device) so the identities are testable against the oracle without a GPU
(tests/test_compose_cpu.py); the engine calls them when it packs weights / builds per-geometry tables.

Reference expressions:
  decoder_input = f @ W2^T + b2                     ffn1's last Linear on its hidden layer f   (networks/zutis.py:500-503)
  K_l = (decoder_input + pos) @ Wk_l^T + bk_l       decoder layer l, cross-attention keys      (networks/transformer.py:281-283)
  V_l =  decoder_input        @ Wv_l^T + bv_l                                                  (networks/transformer.py:284)
  masks = sigmoid(q . decoder_input[m])             mask proposals                             (networks/zutis.py:196-198,209)
  pos[y, x] = [py(y) | px(x)]                       sine positional embedding                  (networks/positional_embedding.py:47-52)
  q = k = tgt + query_pos ; v = tgt                 decoder self-attention inputs              (networks/transformer.py:272-275)
  q_cross = tgt + query_pos                         decoder cross-attention query input        (networks/transformer.py:281-282)
"""
from __future__ import annotations

from typing import Tuple

import torch

f64, f32 = torch.float64, torch.float32


def compose_memory_linear(kw: torch.Tensor, kb: torch.Tensor, vw: torch.Tensor, vb: torch.Tensor, W2: torch.Tensor,
                          b2: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """kw / vw [N, D], kb / vb [N] (all layers concatenated), W2 [D, F], b2 [D] ->
    (Wk W2 [N, F], Wk b2 + bk [N], Wv W2 [N, F], Wv b2 + bv [N]) in fp32:  K = f @ (Wk W2)^T + (Wk b2 + bk) + pos @ Wk^T."""
    kw, kb, vw, vb, W2, b2 = (t.detach().to(f64) for t in (kw, kb, vw, vb, W2, b2))
    return (kw @ W2).to(f32), (kb + kw @ b2).to(f32), (vw @ W2).to(f32), (vb + vw @ b2).to(f32)


def separable_pos_tables(pe: torch.Tensor, wk: torch.Tensor, h: int, w: int, dtype=f32) -> Tuple[torch.Tensor, torch.Tensor]:
    """pe [h*w, D] sine PE (channels [0, D/2) depend on y only, [D/2, D) on x only), wk [N, D] ->
    (Ty [h, N], Tx [w, N]) with (pe @ wk^T)[y*w + x] = Ty[y] + Tx[x]."""
    D = pe.shape[1]
    pe3, wk = pe.detach().view(h, w, D).to(f64), wk.detach().to(f64)
    ty = pe3[:, 0, : D // 2] @ wk[:, : D // 2].t()
    tx = pe3[0, :, D // 2:] @ wk[:, D // 2:].t()
    return ty.to(dtype).contiguous(), tx.to(dtype).contiguous()


def mask_query_weight(W2: torch.Tensor, b2: torch.Tensor, FX: int) -> torch.Tensor:
    """W2 [D, F], b2 [D] -> Wq [FX, D] (FX >= F + 1) with  q . (f @ W2^T + b2) = (Wq q) . [f | 1 | 0...]:
    rows 0..F-1 = W2^T, row F = b2, the rest zero."""
    D, F = W2.shape
    assert FX >= F + 1
    wq = torch.zeros((FX, D), dtype=f32, device=W2.device)
    wq[:F] = W2.detach().t()
    wq[F] = b2.detach()
    return wq


def query_pos_tables(query_pos: torch.Tensor, in_w: torch.Tensor, in_b: torch.Tensor, cross_w: torch.Tensor,
                     cross_b: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Linear layers see `tgt + query_pos` (a parameter, the same Q rows for every image), so
        (tgt + query_pos) @ W^T + b = tgt @ W^T + (query_pos @ W^T + b):
    the projections run on tgt alone and start from a per-query row table.  query_pos [Q, D]; in_w [3D, D], in_b [3D] = the
    packed self-attention in_proj (q | k | v rows; v sees tgt WITHOUT query_pos); cross_w [D, D], cross_b [D] = the q rows of
    the cross-attention in_proj.  Returns (T_self [Q, 3D] = [qp Wq^T + bq | qp Wk^T + bk | bv], T_cross [Q, D]) in fp32."""
    qp, in_w, in_b, cross_w, cross_b = (t.detach().to(f64) for t in (query_pos, in_w, in_b, cross_w, cross_b))
    Q, D = qp.shape
    t_self = torch.cat([qp @ in_w[:2 * D].t() + in_b[:2 * D], in_b[2 * D:].expand(Q, D)], dim=1)
    t_cross = qp @ cross_w.t() + cross_b
    return t_self.to(f32).contiguous(), t_cross.to(f32).contiguous()
