"""The pack-time rewrites of zutis_amd/compose.py against the oracle's own expressions, in fp64 on the CPU:
each identity must hold to rounding of the fp32 storage of the composed weights (no GPU involved)."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import zutis_ref as ref          # noqa: E402
from zutis_amd import compose                # noqa: E402

f64 = torch.float64


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g, dtype=torch.float32) * scale


@pytest.mark.parametrize("h,w", [(6, 10), (7, 5)])
def test_composed_kv_projections_equal_the_reference_expressions(h, w):
    """(decoder_input + pos) @ Wk^T + bk  and  decoder_input @ Wv^T + bv  (transformer.py:281-284, as the oracle's mha() forms
    them from in_proj slices) == the composed weights on ffn1's hidden layer + the separable pos tables."""
    D, Fh, L, B = 48, 16, 3, 2
    M = h * w
    f = F.relu(_rand((B, M, Fh), 1))
    W2, b2 = _rand((D, Fh), 2, 0.2), _rand((D,), 3, 0.1)
    kw, kb, vw, vb = _rand((L * D, D), 4, 0.2), _rand((L * D,), 5, 0.1), _rand((L * D, D), 6, 0.2), _rand((L * D,), 7, 0.1)
    pos = ref.sine_pe(h, w, D)                                               # oracle restatement of positional_embedding.py:29-52
    dec_in = f.to(f64) @ W2.to(f64).t() + b2.to(f64)                         # zutis.py:500-503 (last Linear of ffn1)
    K_ref = (dec_in + pos.to(f64)[None]) @ kw.to(f64).t() + kb.to(f64)
    V_ref = dec_in @ vw.to(f64).t() + vb.to(f64)
    ckw, ckb, cvw, cvb = compose.compose_memory_linear(kw, kb, vw, vb, W2, b2)
    ty, tx = compose.separable_pos_tables(pos, kw, h, w)
    assert ckw.shape == (L * D, Fh) and ty.shape == (h, L * D) and tx.shape == (w, L * D)
    m = torch.arange(M)
    K = f.to(f64) @ ckw.to(f64).t() + ckb.to(f64) + ty.to(f64)[m // w][None] + tx.to(f64)[m % w][None]
    V = f.to(f64) @ cvw.to(f64).t() + cvb.to(f64)
    assert float((K - K_ref).abs().max()) < 2e-6 and float((V - V_ref).abs().max()) < 2e-6


def test_pos_tables_are_exactly_separable():
    """pos @ Wk^T == Ty[y] + Tx[x] needs the sine PE's y-half / x-half channel split (positional_embedding.py:47-52)."""
    h, w, D = 5, 9, 32
    pos = ref.sine_pe(h, w, D).view(h, w, D)
    assert torch.equal(pos[:, :, : D // 2], pos[:, :1, : D // 2].expand(h, w, D // 2))      # y-half constant along x
    assert torch.equal(pos[:, :, D // 2:], pos[:1, :, D // 2:].expand(h, w, D // 2))        # x-half constant along y
    wk = _rand((40, D), 11)
    ty, tx = compose.separable_pos_tables(pos.reshape(h * w, D), wk, h, w)
    full = (pos.reshape(h * w, D).to(f64) @ wk.to(f64).t()).view(h, w, -1)
    assert float((ty.to(f64)[:, None] + tx.to(f64)[None] - full).abs().max()) < 1e-6


def test_mask_einsum_through_the_hidden_layer():
    """sigmoid(q . decoder_input[m]) (zutis.py:196-198,209; oracle zutis_forward) == sigmoid((Wq q) . [f | 1 | 0...][m])."""
    D, Fh, M, Q = 48, 16, 35, 7
    FX = 64
    f = F.relu(_rand((M, Fh), 21))
    W2, b2 = _rand((D, Fh), 22, 0.3), _rand((D,), 23, 0.2)
    q = F.normalize(_rand((Q, D), 24), dim=-1)
    ref_masks = torch.sigmoid(torch.einsum("qc,nc->qn", q.to(f64), f.to(f64) @ W2.to(f64).t() + b2.to(f64)))
    wq = compose.mask_query_weight(W2, b2, FX)
    assert wq.shape == (FX, D) and float(wq[Fh + 1:].abs().max()) == 0.0
    fx = torch.zeros((M, FX), dtype=f64)
    fx[:, :Fh] = f
    fx[:, Fh] = 1
    masks = torch.sigmoid((q.to(f64) @ wq.to(f64).t()) @ fx.t())
    assert float((masks - ref_masks).abs().max()) < 1e-12


def test_first_linear_commutes_with_the_x2_bilinear_upsample():
    """zutis.py:491-503 upsamples the tokens and then applies ffn1's first Linear; the engine applies the Linear first.
    F.interpolate(scale_factor=2, bilinear, align_corners=False) is a convex combination per output pixel, so the two agree."""
    B, h, w, D, N = 2, 5, 7, 24, 10
    tok = _rand((B, h, w, D), 31).to(f64)
    W, b = _rand((N, D), 32).to(f64), _rand((N,), 33).to(f64)
    up = lambda t: F.interpolate(t.permute(0, 3, 1, 2), scale_factor=2, mode="bilinear").permute(0, 2, 3, 1)
    a = F.linear(up(tok), W, b)
    c = up(F.linear(tok, W, b))
    assert float((a - c).abs().max()) < 1e-12


def test_query_pos_row_tables():
    """transformer.py:272-275,281-282: q = k = (tgt + query_pos) W^T + b, v = tgt Wv^T + bv and the cross-attention query
    (tgt + query_pos) Wq^T + bq, for B images of Q queries, == projections of tgt alone + the row tables at row m % Q."""
    B, Q, D = 3, 5, 16
    tgt, qp = _rand((B * Q, D), 41).to(f64), _rand((Q, D), 42)
    in_w, in_b = _rand((3 * D, D), 43, 0.3), _rand((3 * D,), 44, 0.2)
    cw, cb = _rand((D, D), 45, 0.3), _rand((D,), 46, 0.2)
    qin = tgt + qp.to(f64).repeat(B, 1)
    ref_self = torch.cat([F.linear(qin, in_w[:2 * D].to(f64), in_b[:2 * D].to(f64)), F.linear(tgt, in_w[2 * D:].to(f64), in_b[2 * D:].to(f64))], 1)
    ref_cross = F.linear(qin, cw.to(f64), cb.to(f64))
    t_self, t_cross = compose.query_pos_tables(qp, in_w, in_b, cw, cb)
    assert t_self.shape == (Q, 3 * D) and t_cross.shape == (Q, D) and t_self.dtype == torch.float32
    m = torch.arange(B * Q) % Q
    assert float((tgt @ in_w.to(f64).t() + t_self.to(f64)[m] - ref_self).abs().max()) < 1e-6
    assert float((tgt @ cw.to(f64).t() + t_cross.to(f64)[m] - ref_cross).abs().max()) < 1e-6
