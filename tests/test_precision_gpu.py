"""-m gpu: the precision ladder (SURVEY.md §7 "Precision vs roofline").

The reference computes in fp32 end to end (networks/clip_arch.py:286-292, networks/zutis.py:55).  This file pins
 (1) the reference-equivalent contraction mode — fp16 split pairs, three MFMA products (zh_gemm_f16x3, split-pair scores in
     zh_attention_f16) — against float64 references on operands that are NOT fp16-representable;
 (2) the whole forward on the outlier-channel stress model (detgen.stress_state_dict: x100 residual channels, sharpened
     attention, true-fp32 weights) against the fp32 oracle, for both engine precisions:
       "exact" (x3 everywhere): logits / masks within 2e-5 / 2e-4 — fp32-reordering class;
       "fast"  (x3 on the output-facing contractions, fp16 operands in the transformer bodies): within the north-star
               tolerance 1e-3 with margin (asserted at 2.5e-4 / 1e-3).
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

f16, f32, f64 = torch.float16, torch.float32, torch.float64


def _randn(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def _split_act(x32, dev):
    """fp32 [rows, K] -> Act split pair on the device (what a producer kernel with lo_plane != 0 writes)."""
    from zutis_amd.ops import Act
    hi = x32.to(f16)
    lo = (x32 - hi.float()).to(f16)
    return Act(torch.stack([hi, lo]).contiguous().to(dev))


@pytest.mark.parametrize("M,N,K", [(128, 64, 64), (300, 200, 192), (442, 768, 768), (1500, 2304, 768), (2100, 768, 3072),
                                   (100, 1764, 768), (81, 1764, 512), (37, 52, 128), (600, 1, 384)])
def test_gemm_x3_matches_float64(dev, M, N, K):
    """Generic fp32 operands with a x1000 dynamic range: |err| <= 2e-6 * sum_k |a||w| (fp32-class), where the fp16-operand
    GEMM on the same data is ~500x worse."""
    from zutis_amd import ops
    A = _randn((M, K), 1) * torch.exp(_randn((M, K), 11) * 1.5)
    W = _randn((N, K), 2, 0.03) * torch.exp(_randn((N, K), 12) * 1.5)
    bias = _randn((N,), 3)
    ref = A.double() @ W.double().t() + bias.double()
    bound = (A.abs().double() @ W.abs().double().t())
    out = torch.empty((M, N), dtype=f32, device=dev)
    ops.gemm_x3(_split_act(A, dev), ops.split_weight(W.to(dev)), out, bias=bias.to(dev))
    err = (out.cpu().double() - ref).abs()
    assert float((err / (bound + 1e-30)).max()) < 2e-6
    out16 = torch.empty((M, N), dtype=f32, device=dev)
    if K % 64 == 0 and N % 4 == 0:
        ops.gemm(A.to(f16).to(dev), W.to(f16).to(dev), out16, bias=bias.to(dev))
        e16 = (out16.cpu().double() - ref).abs()
        assert float(err.max()) * 50 < float(e16.max())


@pytest.mark.parametrize("act", [0, 1, 2, 3, 4])
def test_gemm_x3_epilogues_and_outputs(dev, act):
    """bias + activation + residual (f32 out), f16 out, and split-pair out whose hi + lo reproduces the fp32 result to 22 bits;
    batched form with a shared A."""
    import torch.nn.functional as F
    from zutis_amd import ops, _lib
    from zutis_amd.ops import Act
    M, N, K = 520, 264, 128
    A, W = _randn((M, K), 3, 0.5), _randn((N, K), 4, 0.2)
    bias, res = _randn((N,), 5), _randn((50, N), 6)
    y = A.double() @ W.double().t() + bias.double()
    y = [y, y * torch.sigmoid(1.702 * y), F.relu(y), torch.sigmoid(y), F.gelu(y)][act]
    Ad, Wd = _split_act(A, dev), ops.split_weight(W.to(dev))
    if act in (0, 3):
        out = torch.empty((M, N), dtype=f32, device=dev)
        ops.gemm_x3(Ad, Wd, out, bias=bias.to(dev), residual=res.to(dev), res_rows=50, act=act)
        ref = y + res.double()[torch.arange(M) % 50]
        assert float((out.cpu().double() - ref).abs().max()) < 3e-6
    if act != 3:
        o16 = Act.empty((M, N), False, dev)
        ops.gemm_x3(Ad, Wd, o16, bias=bias.to(dev), act=act)
        assert torch.allclose(o16.hi.float().cpu().double(), y, atol=4e-3, rtol=1e-3)
        osp = Act.empty((M, N), True, dev)
        ops.gemm_x3(Ad, Wd, osp, bias=bias.to(dev), act=act)
        # the hi plane IS the plain fp16 tensor (the two instantiations may contract the activation differently: <= 1 ulp)
        d = (osp.hi.float() - o16.hi.float()).abs()
        assert float((d / o16.hi.float().abs().clamp_min(6e-5)).max()) <= 2.0 ** -10 and float((d > 0).float().mean()) < 0.01
        both = osp.t[0].float() + osp.t[1].float()
        tol = 2e-6 if act != 4 else 2e-5                                 # erff on the device vs torch's erf
        assert float((both.cpu().double() - y).abs().max()) < tol * max(1.0, float(y.abs().max()))
    else:
        with pytest.raises(_lib.ZutisHipError):
            ops.gemm_x3(Ad, Wd, Act.empty((M, N), False, dev), act=act)


@pytest.mark.parametrize("N", [256, 264])
def test_gemm_x3_pos_tables(dev, N):
    """The composed K projection in the x3 mode: accumulators start from the pos tables (times 2^s, exact); fp32 / fp16 /
    split-pair outputs."""
    from zutis_amd import ops
    from zutis_amd.ops import Act
    hh, ww, B, K = 6, 10, 9, 256
    M = B * hh * ww
    A, W = _randn((M, K), 41, 0.5), _randn((N, K), 42, 0.2)
    bias, Ty, Tx = _randn((N,), 43), _randn((hh, N), 44), _randn((ww, N), 45)
    m = torch.arange(M)
    ref = A.double() @ W.double().t() + bias.double() + Ty.double()[(m % (hh * ww)) // ww] + Tx.double()[m % ww]
    Ad, Wd = _split_act(A, dev), ops.split_weight(W.to(dev))
    kw = dict(bias=bias.to(dev), pos=(Ty.to(dev), Tx.to(dev)))
    o32 = torch.empty((M, N), dtype=f32, device=dev)
    ops.gemm_x3(Ad, Wd, o32, **kw)
    tol = 1e-6 * max(1.0, float(ref.abs().max()))                        # fp32 accumulation that starts from the table value
    assert float((o32.cpu().double() - ref).abs().max()) < tol
    o16 = Act.empty((M, N), False, dev)
    ops.gemm_x3(Ad, Wd, o16, **kw)
    assert torch.equal(o16.hi.cpu(), o32.cpu().to(f16))
    osp = Act.empty((M, N), True, dev)
    ops.gemm_x3(Ad, Wd, osp, **kw)
    assert torch.equal(osp.hi.cpu(), o16.hi.cpu())
    both = osp.t[0].float() + osp.t[1].float()
    assert float((both.cpu().double() - ref).abs().max()) < tol


@pytest.mark.parametrize("tile", [64, 96, 192, 256, 448, 512, 3064])
@pytest.mark.parametrize("hh,ww", [(12, 20), (52, 80)])
def test_gemm_x3_pos_tables_every_tile(dev, tile, hh, ww):
    """The pos tables under every x3 tile, on both table paths: the slice staged through the free ring slot (small images)
    and straight from global memory (slices wider than the slot: 52 x 80 is the decoder memory of a 427 x 640 image)."""
    from zutis_amd import ops, _lib
    from zutis_amd.ops import Act
    L = _lib.load(raw=True)
    N, K, B = 520, 256, 2
    M = B * hh * ww - 5
    A, W = _randn((M, K), 41, 0.5), _randn((N, K), 42, 0.2)
    bias, Ty, Tx = _randn((N,), 43), _randn((hh, N), 44), _randn((ww, N), 45)
    m = torch.arange(M)
    ref = A.double() @ W.double().t() + bias.double() + Ty.double()[(m % (hh * ww)) // ww] + Tx.double()[m % ww]
    Ad, Wd = _split_act(A, dev), ops.split_weight(W.to(dev))
    _lib.check(L.zh_dev_set_gemm_overrides(0, tile, 0), "zh_dev_set_gemm_overrides")
    try:
        osp = Act.empty((M, N), True, dev)
        ops.gemm_x3(Ad, Wd, osp, bias=bias.to(dev), pos=(Ty.to(dev), Tx.to(dev)))
        both = osp.t[0].float() + osp.t[1].float()
        err = float((both.cpu().double() - ref).abs().max())
        assert err < 1e-6 * max(1.0, float(ref.abs().max())), (tile, hh, ww, err)
    finally:
        L.zh_dev_set_gemm_overrides(0, 0, 0)


def test_gemm_x3_batched_activation_operands(dev):
    """The mask einsum form (zutis.py:196-198): both operands are activations (split pairs, scale 1), batched."""
    from zutis_amd import ops
    B, Q, Mpix, D = 3, 100, 1764, 768
    q, t = _randn((B * Q, D), 1, 0.05), _randn((B * Mpix, D), 2, 2.0)
    out = torch.empty((B, Q, Mpix), dtype=f32, device=dev)
    ops.gemm_x3(_split_act(q, dev), _split_act(t, dev), out, act=ops.ACT_SIGMOID, M=Q, N=Mpix, K=D, lda=D, ldw=D, ldc=Mpix,
                batch=B, strideA=Q * D, strideW=Mpix * D, strideC=Q * Mpix)
    ref = torch.sigmoid(torch.einsum("bqc,bnc->bqn", q.view(B, Q, D).double(), t.view(B, Mpix, D).double()))
    assert float((out.cpu().double() - ref).abs().max()) < 2e-6


def test_gemm_x3_race_screen_bitwise_repeatable(dev):
    """The x3 K loop is a 3-slot LDS-DMA ring with counted vmcnt waits: a mis-counted wait shows up as rare wrong tiles."""
    from zutis_amd import ops
    # M >= 4096 reaches the two-slot 256x256 / 192x256 tiles (barrier + LDS-DMA into the slot just read: lgkmcnt(0) before it)
    for (M, N, K) in [(3000, 2304, 768), (1500, 768, 3072), (700, 640, 192), (300, 200, 64), (5000, 512, 128), (4500, 2304, 768),
                      (4200, 768, 3072), (4100, 4608, 256), (3200, 768, 768)]:
        A, W = _split_act(_randn((M, K), 100 + M), dev), ops.split_weight(_randn((N, K), 200 + N, 0.05).to(dev))
        outs = []
        for _ in range(12):
            o = torch.empty((M, N), dtype=f32, device=dev)
            ops.gemm_x3(A, W, o)
            outs.append(o)
        torch.cuda.synchronize()
        for o in outs[1:]:
            assert torch.equal(o, outs[0])
        ref = (A.t[0].double() + A.t[1].double()) @ ((W.t[0].double() + W.t[1].double()) * W.out_scale).t()
        assert float((outs[0].double() - ref).abs().max()) < 1e-4 * math.sqrt(K / 64)


@pytest.mark.parametrize("kind", ["split", "f16", "f32"])
def test_gemm_x3_row_periodic_table_before_rounding(dev, kind):
    """The decoder's query_pos terms (transformer.py:272-275,281-282 by linearity): out[m] = A[m] W^T + table[m % Q] with a table
    that DWARFS the products (queries x20: |table| ~ 500 against products of O(1)).  The table must join the finished fp32
    accumulator once (fp32-class: |err| <= a few ulp of the table), for f32, fp16 and split-pair outputs."""
    from zutis_amd import ops
    from zutis_amd.ops import Act
    B, Q, N, K = 5, 100, 2304, 768
    M = B * Q
    A32, W32 = _randn((M, K), 1), _randn((N, K), 2, 0.03)
    T = _randn((Q, N), 3, 150.0)
    ref = A32.double() @ W32.double().t() + T.double().repeat(B, 1)
    A, W = _split_act(A32, dev), ops.split_weight(W32.to(dev))
    if kind == "f32":
        out = torch.empty((M, N), dtype=f32, device=dev)
        ops.gemm_x3(A, W, out, residual=T.to(dev), res_rows=Q)
        got, tol = out.cpu().double(), 6e-5                      # one fp32 rounding at |out| <= ~700: ulp 6e-5
    elif kind == "split":
        out = Act.empty((M, N), True, dev)
        ops.gemm_x3(A, W, out, residual=T.to(dev), res_rows=Q)
        got, tol = out.t[0].float().cpu().double() + out.t[1].float().cpu().double(), 2.5e-4   # 22 bits of ~700
    else:
        out = Act.empty((M, N), False, dev)
        ops.gemm_x3(A, W, out, residual=T.to(dev), res_rows=Q)
        got, tol = out.hi.float().cpu().double(), 0.3            # fp16 of ~700: ulp 0.5
        assert torch.equal(out.hi.cpu(), ref.to(torch.float32).to(f16)) or float((got - ref).abs().max()) <= 0.5
    assert float((got - ref).abs().max()) < tol, float((got - ref).abs().max())


@pytest.mark.parametrize("tile", [64, 96, 192, 256, 512, 448, 3064, 3066, 32, 1288, 6496, 6464, 7096, 7128])
def test_gemm_x3_every_tile_variant(dev, tile):
    """Every x3 tile (128x64, 192x128, 256x128 on the 3-slot ring; 256x256 on the two-slot ring with the SGPR-base LDS-DMA and
    the in-place A lo fragments), forced through zh_dev_set_gemm_overrides, over K = 64 .. 1024 (every prologue / steady / tail
    phase), ragged M and N, f32 / split-pair outputs with bias, activation and residual: fp32-class against float64 and bitwise
    repeatable."""
    from zutis_amd import ops, _lib
    from zutis_amd.ops import Act
    L = _lib.load(raw=True)
    _lib.check(L.zh_dev_set_gemm_overrides(0, tile, 0), "zh_dev_set_gemm_overrides")
    try:
        for (M, N) in ((333, 328), (700, 520)):
            for K in (64, 128, 192, 256, 320, 448, 512, 1024):
                A32, W32 = _randn((M, K), 300 + K, 0.5), _randn((N, K), 400 + K, 0.05)
                bias, res = _randn((N,), 5), _randn((M, N), 6)
                A, W = _split_act(A32, dev), ops.split_weight(W32.to(dev))
                ref = A32.double() @ W32.double().t() + bias.double()
                bound = float((A32.abs().double() @ W32.abs().double().t()).max())
                outs = []
                for _ in range(3):
                    o = torch.empty((M, N), dtype=f32, device=dev)
                    ops.gemm_x3(A, W, o, bias=bias.to(dev), residual=res.to(dev))
                    outs.append(o)
                assert float((outs[0].cpu().double() - (ref + res.double())).abs().max()) < 2e-6 * bound + 1e-6, (tile, M, N, K)
                assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (tile, M, N, K)
                sp = Act.empty((M, N), True, dev)
                ops.gemm_x3(A, W, sp, bias=bias.to(dev), act=ops.ACT_RELU)
                got = sp.t[0].float().cpu().double() + sp.t[1].float().cpu().double()
                want = torch.relu(ref)
                assert float((got - want).abs().max()) < 2e-6 * bound + 2e-6 * float(want.abs().max()) + 1e-6, (tile, M, N, K)
    finally:
        L.zh_dev_set_gemm_overrides(0, 0, 0)


def test_split_producers_write_hi_plus_lo(dev):
    """Every producer with a lo_plane argument: hi plane == the plain fp16 output, hi + lo == the fp32 value to ~2^-22."""
    from zutis_amd import ops
    from zutis_amd.ops import Act
    rows, D = 300, 768
    x = (_randn((rows, D), 1) * torch.exp(_randn((rows, D), 2))).to(dev)
    g, b = _randn((D,), 3, 0.1).add(1).to(dev), _randn((D,), 4, 0.1).to(dev)

    def check(pair: Act, plain: Act, ref32, tol=3e-7):
        assert torch.equal(pair.hi, plain.hi)
        both = pair.t[0].float() + pair.t[1].float()
        scale = ref32.abs().clamp_min(0.25)
        assert float(((both - ref32).abs() / scale).max()) < tol

    y32 = torch.empty((rows, D), dtype=f32, device=dev)
    sp, pl = Act.empty((rows, D), True, dev), Act.empty((rows, D), False, dev)
    ops.layernorm(x, g, b, 1e-5, rows, D, out_f32=y32, out_f16=sp)
    ops.layernorm(x, g, b, 1e-5, rows, D, out_f16=pl)
    check(sp, pl, y32)
    ops.cast_f16(x, sp, rows, D); ops.cast_f16(x, pl, rows, D)
    check(sp, pl, x)
    ops.l2norm_rows(x, rows, D, out_f32=y32, out_f16=sp); ops.l2norm_rows(x, rows, D, out_f16=pl)
    check(sp, pl, y32)
    B, h, w = 2, 5, 7
    t = _randn((B, h * w, D), 5).to(dev)
    u32 = torch.empty((B * 4 * h * w, D), dtype=f32, device=dev)
    usp, upl = Act.empty((B * 4 * h * w, D), True, dev), Act.empty((B * 4 * h * w, D), False, dev)
    ops.upsample2x_cl(t, B, h, w, D, out_f32=u32, out_f16=usp); ops.upsample2x_cl(t, B, h, w, D, out_f16=upl)
    check(usp, upl, u32)
    pe = _randn((4 * h * w, D), 6).to(dev)
    ksp, kpl = Act.empty((B * 4 * h * w, D), True, dev), Act.empty((B * 4 * h * w, D), False, dev)
    ops.add_rowperiodic_f16(usp, pe, ksp, B * 4 * h * w, D, 4 * h * w)
    ref = (usp.t[0].float() + usp.t[1].float()) + pe.repeat(B, 1)
    both = ksp.t[0].float() + ksp.t[1].float()
    assert float(((both - ref).abs() / ref.abs().clamp_min(0.25)).max()) < 3e-7
    ops.add_rowperiodic_f16(upl, pe, kpl, B * 4 * h * w, D, 4 * h * w)
    assert torch.equal(kpl.hi, (upl.hi.float() + pe.repeat(B, 1)).to(f16))
    # im2col: scalar and vector paths, zero padding of K
    img = _randn((2, 3, 80, 112), 7).to(dev)
    Kp = 768
    csp, cpl = Act.empty((2 * 5 * 7, Kp), True, dev), Act.empty((2 * 5 * 7, Kp), False, dev)
    ops.im2col(img, csp, 16, Kp); ops.im2col(img, cpl, 16, Kp)
    cols = torch.nn.functional.unfold(img, 16, stride=16).transpose(1, 2).reshape(-1, 768)
    check(csp, cpl, cols)
    img2 = _randn((1, 3, 37, 45), 8).to(dev)
    c2 = Act.empty((5 * 6, 192), True, dev)
    ops.im2col(img2, c2, 8, 192, pad_to_patch=True)
    pad = torch.nn.functional.pad(img2, (0, 3, 0, 3))
    cols2 = torch.nn.functional.unfold(pad, 8, stride=8).transpose(1, 2).reshape(-1, 192)
    assert float(((c2.t[0].float() + c2.t[1].float()) - cols2).abs().max()) < 1e-6
    # global LN + L2
    ts = _randn((2, 4 * h * w, 512), 9).to(dev).contiguous()
    p32 = torch.empty_like(ts)
    psp, ppl = Act.empty((2 * 4 * h * w, 512), True, dev), Act.empty((2 * 4 * h * w, 512), False, dev)
    ops.global_ln_l2(ts, 2, 4 * h * w, 512, out_f32=p32, out_f16=psp); ops.global_ln_l2(ts, 2, 4 * h * w, 512, out_f16=ppl)
    assert torch.equal(psp.hi, ppl.hi)
    assert float(((psp.t[0].float() + psp.t[1].float()).view_as(p32) - p32).abs().max()) < 1e-7


@pytest.mark.parametrize("dh,heads,Tq,Tk,causal", [(64, 3, 442, 442, False), (96, 2, 100, 1764, False), (64, 2, 77, 77, True), (64, 1, 130, 700, False)])
def test_attention_x3_scores(dev, dh, heads, Tq, Tk, causal):
    """Split-pair attention: large, sharply peaked logits (|s| up to ~60) where fp16 operand rounding moves the softmax by
    ~1e-2; the x3 form (split scores AND split P.V) stays at fp32-class error against float64."""
    from zutis_amd import ops
    from zutis_amd.ops import Act
    B, D = 2, heads * dh
    q, k, v = _randn((B * Tq, D), 1, 2.5), _randn((B * Tk, D), 2, 2.5), _randn((B * Tk, D), 3)
    qd, kd = q.view(B, Tq, heads, dh).transpose(1, 2).double(), k.view(B, Tk, heads, dh).transpose(1, 2).double()
    vd = v.view(B, Tk, heads, dh).transpose(1, 2).double()
    s = qd @ kd.transpose(-1, -2) / math.sqrt(dh)
    if causal:
        s = s + torch.full((Tq, Tk), float("-inf"), dtype=f64).triu_(1)
    ref = (torch.softmax(s, -1) @ vd).transpose(1, 2).reshape(B * Tq, D)
    Q, K, V = _split_act(q, dev), _split_act(k, dev), _split_act(v, dev)
    kw = dict(batch=B, heads=heads, Tq=Tq, Tk=Tk, head_dim=dh, ldq=D, ldk=D, ldv=D, ldo=D, strideQ=Tq * D, strideK=Tk * D,
              strideV=Tk * D, strideO=Tq * D, causal=causal)
    O3, O1 = Act.empty((B * Tq, D), True, dev), Act.empty((B * Tq, D), False, dev)
    ops.attention(Q, K, V, O3, x3=True, **kw)
    ops.attention(Q, K, V, O1, x3=False, **kw)
    e3 = float(((O3.t[0].float() + O3.t[1].float()).cpu().double() - ref).abs().max())
    e1 = float((O1.hi.float().cpu().double() - ref).abs().max())
    assert e3 < 2e-5 and e3 * 100 < e1, (e3, e1)


@pytest.mark.parametrize("x3", [True, False])
@pytest.mark.parametrize("dh,heads,Tq,Tk,S", [(96, 8, 100, 1764, 2), (96, 2, 100, 1764, 3), (64, 3, 130, 1000, 4), (96, 1, 20, 5504, 8)])
def test_attention_key_split_matches_unsplit(dev, dh, heads, Tq, Tk, S, x3):
    """zh_attention_f16_splitk (keys split over S workgroups + merge launch) against the single-pass kernel and float64: the
    merge is the online-softmax rescale, so the result differs from the unsplit one only by fp32 reassociation."""
    from zutis_amd import ops
    from zutis_amd.ops import Act
    B, D = 3, heads * dh
    q, k, v = _randn((B * Tq, D), 1, 1.5), _randn((B * Tk, D), 2, 1.5), _randn((B * Tk, D), 3)
    qd, kd = q.view(B, Tq, heads, dh).transpose(1, 2).double(), k.view(B, Tk, heads, dh).transpose(1, 2).double()
    vd = v.view(B, Tk, heads, dh).transpose(1, 2).double()
    ref = (torch.softmax(qd @ kd.transpose(-1, -2) / math.sqrt(dh), -1) @ vd).transpose(1, 2).reshape(B * Tq, D)
    mk = (lambda t: _split_act(t, dev)) if x3 else (lambda t: Act(t.to(f16)[None].contiguous().to(dev)))
    Q, K, V = mk(q), mk(k), mk(v)
    kw = dict(batch=B, heads=heads, Tq=Tq, Tk=Tk, head_dim=dh, ldq=D, ldk=D, ldv=D, ldo=D, strideQ=Tq * D, strideK=Tk * D,
              strideV=Tk * D, strideO=Tq * D, x3=x3)
    O1, O2 = Act.empty((B * Tq, D), x3, dev), Act.empty((B * Tq, D), x3, dev)
    ops.attention(Q, K, V, O1, **kw)
    ws = torch.empty(ops.attention_splitk_workspace_size(B, heads, Tq, dh, S), dtype=torch.uint8, device=dev)
    ops.attention(Q, K, V, O2, ksplit=S, workspace=ws, **kw)
    val = (lambda o: (o.t[0].float() + o.t[1].float()) if x3 else o.hi.float())
    e1 = float((val(O1).cpu().double() - ref).abs().max())
    e2 = float((val(O2).cpu().double() - ref).abs().max())
    if x3:
        assert e2 < 2e-5, (e1, e2)
    else:      # same operand rounding as the single pass: within the same error of float64
        assert e2 < 2 * e1 + 1e-3, (e1, e2)
    assert float((val(O1) - val(O2)).abs().max()) < (1e-5 if x3 else 2e-3)


@pytest.mark.parametrize("T", [442, 449, 64])
def test_attention_x3_pipelined_loop_is_bitwise_the_plain_loop(dev, T):
    """The split-pair dh = 64 kernel runs its software-pipelined loop on small grids and the plain loop on large ones (the
    launcher's rule in attention.hip): image 0 alone (48 workgroups at T = 442: pipelined) and as the first of 40 images (1920
    workgroups: plain) must agree bit for bit — the two loops issue the same arithmetic, only in a different order in time — and
    both must match float64.  T = 449 = 14 tiles + 1 key (a ragged peeled tile), T = 64 = two full tiles (nothing ragged)."""
    from zutis_amd import ops
    from zutis_amd.ops import Act
    dh, heads, B = 64, 12, 40
    D = heads * dh
    q, k, v = _randn((B * T, D), 11, 1.5), _randn((B * T, D), 12, 1.5), _randn((B * T, D), 13)
    Q, K, V = _split_act(q, dev), _split_act(k, dev), _split_act(v, dev)
    kw = dict(heads=heads, Tq=T, Tk=T, head_dim=dh, ldq=D, ldk=D, ldv=D, ldo=D, strideQ=T * D, strideK=T * D, strideV=T * D, strideO=T * D, x3=True)
    O_all, O_one = Act.empty((B * T, D), True, dev), Act.empty((B * T, D), True, dev)
    O_one.t.zero_()
    ops.attention(Q, K, V, O_all, batch=B, **kw)
    ops.attention(Q, K, V, O_one, batch=1, **kw)
    assert torch.equal(O_all.t[:, :T], O_one.t[:, :T])
    qd, kd, vd = (t[:T].view(T, heads, dh).transpose(0, 1).double() for t in (q, k, v))
    ref = (torch.softmax(qd @ kd.transpose(-1, -2) / math.sqrt(dh), -1) @ vd).transpose(0, 1).reshape(T, D)
    got = (O_one.t[0, :T].float() + O_one.t[1, :T].float()).cpu().double()
    assert float((got - ref).abs().max()) < 2e-5


@pytest.mark.parametrize("dh,heads,B,Tq,Tk", [(96, 8, 4, 100, 1764), (64, 12, 1, 449, 449), (64, 6, 1, 2200, 2200), (96, 8, 2, 100, 70)])
def test_attention_x3_pipelined_loop_race_screen(dev, dh, heads, B, Tq, Tk):
    """The pipelined split-pair loop overlaps LDS tile stores of K(t+2) / V(t+1) with reads of K(t+1) / V(t) behind ONE barrier per
    tile (+ one after the first K.Q^T): 40 repeats of the same launch on grids that take that loop must return the same bits
    (a missing barrier showed up as run-to-run differences on the first implementation)."""
    from zutis_amd import ops
    from zutis_amd.ops import Act
    D = heads * dh
    Q, K, V = (_split_act(_randn((B * T, D), 21 + i, 1.2), dev) for i, T in enumerate((Tq, Tk, Tk)))
    kw = dict(batch=B, heads=heads, Tq=Tq, Tk=Tk, head_dim=dh, ldq=D, ldk=D, ldv=D, ldo=D, strideQ=Tq * D, strideK=Tk * D,
              strideV=Tk * D, strideO=Tq * D, x3=True)
    O0 = Act.empty((B * Tq, D), True, dev)
    ops.attention(Q, K, V, O0, **kw)
    ref = O0.t.clone()
    O = Act.empty((B * Tq, D), True, dev)
    for _ in range(40):
        O.t.fill_(7.0)
        ops.attention(Q, K, V, O, **kw)
        assert torch.equal(O.t, ref)


@pytest.mark.parametrize("x3", [True, False])
@pytest.mark.parametrize("per_tile", [3.0, 7.9, 8.1, 20.0])
def test_attention_lazy_running_max_on_rising_scores(dev, x3, per_tile):
    """The flash kernels move their softmax reference point only when a tile's maximum exceeds it by more than 2^8
    (attention.hip: lazy running max).  Keys whose scores RISE by `per_tile` log2 units from one key tile to the next keep the
    reference stale for as long as the rule allows (3.0: two tiles in three; 7.9: every other tile, probabilities up to 2^7.9;
    8.1 / 20: every tile) — the result must stay the float64 softmax to the precision of the mode."""
    from zutis_amd import ops
    from zutis_amd.ops import Act
    dh, heads, B, T = 64, 2, 2, 448
    D = heads * dh
    kt = 32 if x3 else 64
    g = torch.Generator().manual_seed(5)
    u = torch.nn.functional.normalize(torch.randn(heads, dh, generator=g), dim=-1)              # one direction per head
    ramp = (torch.arange(T) // kt).float() * per_tile * math.log(2.0) * math.sqrt(dh)           # score offset of key j (natural log units)
    q = (0.3 * torch.randn(B, T, heads, dh, generator=g) + u).reshape(B * T, D)
    k = (0.3 * torch.randn(B, T, heads, dh, generator=g) + ramp.view(1, T, 1, 1) * u / (1.0 + 0.0)).reshape(B * T, D)
    v = torch.randn(B * T, D, generator=g)
    if not x3:                                                                                  # the fp16 kernel sees fp16 operands
        q, k, v = q.half().float(), k.half().float(), v.half().float()
    qd, kd, vd = (t.view(B, T, heads, dh).transpose(1, 2).double() for t in (q, k, v))
    ref = (torch.softmax(qd @ kd.transpose(-1, -2) / math.sqrt(dh), -1) @ vd).transpose(1, 2).reshape(B * T, D)
    mk = (lambda t: _split_act(t, dev)) if x3 else (lambda t: Act(t.to(f16)[None].contiguous().to(dev)))
    O = Act.empty((B * T, D), x3, dev)
    ops.attention(mk(q), mk(k), mk(v), O, batch=B, heads=heads, Tq=T, Tk=T, head_dim=dh, ldq=D, ldk=D, ldv=D, ldo=D,
                  strideQ=T * D, strideK=T * D, strideV=T * D, strideO=T * D, x3=x3)
    got = ((O.t[0].float() + O.t[1].float()) if x3 else O.hi.float()).cpu().double()
    assert torch.isfinite(got).all()
    err = float((got - ref).abs().max())
    assert err < (3e-5 if x3 else 3e-3), err


# ------------------------------------------------------------------------------------------- whole-model stress test
def _stress_case(dev, B, S, sharp=3.0):
    from zutis_amd import detgen
    from oracle import zutis_ref as O
    cfg = detgen.VIT_B16
    sd = detgen.stress_state_dict(cfg, 100.0)
    D = cfg.width
    for i in range(cfg.layers):                    # sharpen the encoder's attention: |scores| up to ~50 instead of ~6
        sd[f"encoder.transformer.resblocks.{i}.attn.in_proj_weight"][:2 * D] *= np.float32(sharp)
    x = torch.from_numpy(detgen.images(B, S, S))
    text = torch.from_numpy(detgen.text_embeddings(81, cfg.embed_dim))
    with torch.no_grad():
        ref = O.zutis_forward(O.to_torch_params(sd), x, cfg.patch, cfg.dec_heads)
        lo_ref = O.semantic_logits_lowres(ref["patch_tokens"], text).numpy()
        lab_ref = O.predict_semantic(ref["patch_tokens"], text, size=(S, S))
    return cfg, sd, x, text, ref, lo_ref, lab_ref


@pytest.fixture(scope="module")
def stress(dev):
    return _stress_case(dev, 2, 336)


@pytest.mark.parametrize("precision,tol_logit,tol_mask", [("exact", 2e-5, 2e-4), ("fast", 2.5e-4, 1e-3)])
def test_stress_model_c2(dev, stress, precision, tol_logit, tol_mask):
    """ViT-B/16 @336 (config 2 geometry), x100 outlier residual channels, sharpened attention, generic fp32 weights:
    HIP engine vs the fp32 oracle.  North-star tolerance: logits within 1e-3."""
    from zutis_amd.engine import ZutisEngine
    cfg, sd, x, text, ref, lo_ref, lab_ref = stress
    eng = ZutisEngine({k: torch.from_numpy(v).to(dev) for k, v in sd.items()}, cfg.patch, cfg.dec_heads, precision=precision)
    out = eng.forward(x.to(dev))
    lo = eng.semantic_logits_lowres(out["patch_tokens"], text.to(dev)).cpu().numpy()
    lab = eng.predict_semantic(out["patch_tokens"], text.to(dev), (336, 336)).cpu().numpy()
    e_lo = float(np.abs(lo - lo_ref).max())
    e_m = float((out["mask_proposals"].cpu() - ref["mask_proposals"]).abs().max())
    e_pt = float((out["patch_tokens"].cpu() - ref["patch_tokens"]).abs().max())
    agree = float((lab == lab_ref).mean())
    from oracle.parity import unexplained_label_mismatches
    n_mis, n_bad, worst = unexplained_label_mismatches(lab, lab_ref, lo_ref, e_lo, (336, 336))
    print(f"stress[{precision}]: logits {e_lo:.2e} masks {e_m:.2e} patch_tokens {e_pt:.2e} labels {agree:.6f} "
          f"({n_mis} differ, {n_bad} unexplained, largest reference margin {worst:.2e})")
    assert e_lo < tol_logit and e_pt < tol_logit and e_m < tol_mask, (e_lo, e_pt, e_m)
    assert n_bad == 0, (n_mis, n_bad, worst, e_lo)       # every differing pixel sits on a reference top-2 margin <= 2 x logit error


# ---- f16x2: the planeW = 0 form of zh_gemm_f16x3 for fp16-valued weights (the released CLIP towers after the reference's
#      convert_weights, clip_arch.py:566-587,625): the product with the zero lo plane is skipped, nothing else changes
def _f16_valued(shape, seed, scale):
    return (_randn(shape, seed, scale).to(f16)).float()


def test_split_weight_packs_fp16_valued_weights_as_one_plane(dev):
    from zutis_amd import ops
    W16 = _f16_valued((96, 128), 7, 0.05).to(dev)
    a = ops.split_weight(W16)
    assert a.x2 and a.plane == 0 and a.t.shape[0] == 1
    assert torch.equal(a.hi.float() * a.out_scale, W16)                 # the one plane IS the weight, exactly
    b = ops.split_weight(W16, allow_x2=False)
    assert (not b.x2) and b.plane != 0 and not bool(b.t[1].any()) and torch.equal(b.t[0], a.hi)
    c = ops.split_weight(_randn((96, 128), 8, 0.05).to(dev))             # generic fp32 values keep both planes
    assert (not c.x2) and c.plane != 0 and bool(c.t[1].any())
    # fp16 subnormals scale up to normal numbers: still exact, still one plane
    Wd = (torch.tensor([[2.0 ** -24, 3 * 2.0 ** -24, 2.0 ** -15, 0.5]] * 8).repeat(1, 16)).to(dev)
    d = ops.split_weight(Wd)
    assert d.x2 and torch.equal(d.hi.float() * d.out_scale, Wd)
    # a plain fp16 activation is NOT accepted as the weight operand of the x3 entry (its lo plane was never looked at)
    with pytest.raises(Exception):
        ops.gemm_x3(_split_act(_randn((64, 128), 1), dev), ops.Act(W16.to(f16).unsqueeze(0).contiguous()), torch.empty((64, 96), dtype=f32, device=dev))


@pytest.mark.parametrize("tile", [0, 64, 96, 192, 256, 512, 448, 3064, 5122, 5124, 4484, 1288, 32, 6496, 3066, 6464])
def test_gemm_x2_is_bitwise_the_x3_kernel_on_fp16_valued_weights(dev, tile):
    """Every tile of the two-product kernel (incl. the three-slot big tiles and their two-slot A/B form) over every K phase,
    ragged M / N, bias + residual (f32 out), ReLU / QuickGELU split-pair out, fp16 out: bit-identical to the three-product
    kernel fed the same weight with an explicit zero lo plane, and fp32-class against float64."""
    from zutis_amd import ops, _lib
    from zutis_amd.ops import Act
    L = _lib.load(raw=True)
    try:
        for (M, N) in ((333, 328), (700, 520), (4500, 776)):
            for K in ((64, 128, 192, 256, 320, 448, 512, 1024) if M < 4000 else (64, 128, 192, 768, 1024)):
                A32, W32 = _randn((M, K), 300 + K, 0.5), _f16_valued((N, K), 400 + K, 0.05)
                bias, res = _randn((N,), 5), _randn((M, N), 6)
                A = _split_act(A32, dev)
                W2, W3 = ops.split_weight(W32.to(dev)), ops.split_weight(W32.to(dev), allow_x2=False)
                assert W2.x2 and not W3.x2
                ref = A32.double() @ W32.double().t() + bias.double()
                bound = float((A32.abs().double() @ W32.abs().double().t()).max())

                def run(W, t):
                    _lib.check(L.zh_dev_set_gemm_overrides(0, t, 0), "zh_dev_set_gemm_overrides")
                    o = torch.empty((M, N), dtype=f32, device=dev)
                    ops.gemm_x3(A, W, o, bias=bias.to(dev), residual=res.to(dev))
                    sp = Act.empty((M, N), True, dev)
                    ops.gemm_x3(A, W, sp, bias=bias.to(dev), act=ops.ACT_QUICKGELU if K % 128 else ops.ACT_RELU)
                    h = Act.empty((M, N), False, dev)
                    ops.gemm_x3(A, W, h, bias=bias.to(dev))
                    return o, sp.t.clone(), h.t.clone()
                o2, sp2, h2 = run(W2, tile)
                o2b, _, _ = run(W2, tile)
                o3, sp3, h3 = run(W3, {5122: 512, 5124: 512, 4484: 448}.get(tile, tile))     # 5122 / 5124 / 4484 exist for the x2 form only
                assert torch.equal(o2, o3) and torch.equal(sp2, sp3) and torch.equal(h2, h3), (tile, M, N, K)
                assert torch.equal(o2, o2b), (tile, M, N, K)
                assert float((o2.cpu().double() - (ref + res.double())).abs().max()) < 2e-6 * bound + 1e-6, (tile, M, N, K)
    finally:
        L.zh_dev_set_gemm_overrides(0, 0, 0)


def test_gemm_x2_race_screen_and_pos_tables(dev):
    """The three-slot big-tile loop of the x2 kernel (counted vmcnt(NP) with one stage in flight, run-time slot index): bitwise
    repeatable over 12 launches at the model's big shapes, and equal to the x3 kernel; pos tables on both table paths."""
    from zutis_amd import ops
    for (M, N, K) in [(4500, 2304, 768), (4200, 768, 3072), (4100, 4608, 256), (5000, 1024, 1024), (4300, 3072, 1024), (8200, 1024, 4096)]:
        A = _split_act(_randn((M, K), 100 + M), dev)
        W32 = _f16_valued((N, K), 200 + N, 0.05).to(dev)
        W2, W3 = ops.split_weight(W32), ops.split_weight(W32, allow_x2=False)
        outs = []
        for _ in range(12):
            o = torch.empty((M, N), dtype=f32, device=dev)
            ops.gemm_x3(A, W2, o)
            outs.append(o)
        o3 = torch.empty((M, N), dtype=f32, device=dev)
        ops.gemm_x3(A, W3, o3)
        torch.cuda.synchronize()
        for o in outs:
            assert torch.equal(o, o3), (M, N, K)
    for (hh, ww, N) in ((42, 42, 512), (20, 300, 768)):                   # LDS-staged slice / direct path (wide image)
        B, K = 3, 256
        M = B * hh * ww
        A = _split_act(_randn((M, K), 9), dev)
        W32 = _f16_valued((N, K), 10, 0.05).to(dev)
        ty, tx = _randn((hh, N), 11).to(dev), _randn((ww, N), 12).to(dev)
        o2, o3 = torch.empty((M, N), dtype=f32, device=dev), torch.empty((M, N), dtype=f32, device=dev)
        ops.gemm_x3(A, ops.split_weight(W32), o2, pos=(ty, tx))
        ops.gemm_x3(A, ops.split_weight(W32, allow_x2=False), o3, pos=(ty, tx))
        assert torch.equal(o2, o3)
        pos = (ty[:, None, :] + tx[None, :, :]).reshape(hh * ww, N).repeat(B, 1)
        ref = (A.t[0].double() + A.t[1].double()) @ W32.double().t() + pos.double()
        assert float((o2.double() - ref).abs().max()) < 1e-4


@pytest.mark.parametrize("x2", [False, True])
def test_gemm_x3_tail_peel_is_bitwise_one_launch(dev, x2):
    """A big-tile GEMM a few tiles over whole rounds of the chip (257 m-tiles x 4 n-tiles = 4 rounds + 4 tiles) is run as the
    whole rounds + a small-tile call on the last m-tile row (gemm_x3.hip "tail peel"): bitwise the single forced-tile launch, for
    the fp32 + residual and the split-pair epilogues, ragged last tile included."""
    from zutis_amd import ops, _lib
    from zutis_amd.ops import Act
    L = _lib.load(raw=True)
    M, N, K = 257 * 256 - 37, 1024, 128
    A = _split_act(_randn((M, K), 1, 0.5), dev)
    W32 = (_f16_valued if x2 else _randn)((N, K), 2, 0.05).to(dev)
    W = ops.split_weight(W32)
    assert W.x2 == x2
    bias, res = _randn((N,), 3).to(dev), _randn((M, N), 4).to(dev)

    def run():
        o = torch.empty((M, N), dtype=f32, device=dev)
        ops.gemm_x3(A, W, o, bias=bias, residual=res)
        sp = Act.empty((M, N), True, dev)
        ops.gemm_x3(A, W, sp, bias=bias, act=ops.ACT_QUICKGELU)
        return o, sp.t
    o_p, sp_p = run()                                   # cost model: 256 x 256 tiles, 1028 of them -> peeled
    try:
        _lib.check(L.zh_dev_set_gemm_overrides(0, 512, 0), "zh_dev_set_gemm_overrides")
        o_1, sp_1 = run()                               # forced tile: one launch, no peel
    finally:
        L.zh_dev_set_gemm_overrides(0, 0, 0)
    assert torch.equal(o_p, o_1) and torch.equal(sp_p, sp_1)
    ref = (A.t[0].double() + A.t[1].double()) @ W32.double().t() + bias.double() + res.double()
    assert float((o_p.double() - ref).abs().max()) < 1e-4


@pytest.mark.parametrize("M,N,K,batch", [(100, 768, 768, 1), (100, 2304, 768, 1), (100, 768, 2048, 1), (20, 384, 384, 1), (7, 8, 64, 1), (128, 260, 320, 3), (97, 36, 1024, 2), (5, 30, 64, 1), (33, 1, 128, 1)])
def test_gemm_x3_few_row_kernel(dev, M, N, K, batch):
    """The few-row kernel behind zh_gemm_f16x3 (M <= 128: operands straight into fragments, K split over the four waves of a block,
    gemm_skinny.h) on the decoder's batch-1 shapes and ragged ones: fp32-class against float64 for every output kind and epilogue
    piece, the fp16-valued-weight (two-product) form bitwise equal to the three-product form on a zero lo plane, repeatable."""
    from zutis_amd import ops
    from zutis_amd.ops import Act
    A32 = _randn((batch, M, K), 70, 0.7) * torch.exp(_randn((batch, M, K), 71) * 1.2)
    W32 = _randn((batch, N, K), 72, 0.04) * torch.exp(_randn((batch, N, K), 73))
    bias, res = _randn((N,), 74), _randn((25, N), 75)
    A = Act(torch.stack([A32.to(f16), (A32 - A32.to(f16).float()).to(f16)]).contiguous().to(dev))
    Wt = ops.split_weight(W32.to(dev) if batch > 1 else W32[0].to(dev))          # one scale for the whole (batched) weight
    ref = torch.einsum("bmk,bnk->bmn", A32.double(), W32.double())
    bound = torch.einsum("bmk,bnk->bmn", A32.abs().double(), W32.abs().double())
    kw = dict(M=M, N=N, K=K, lda=K, ldw=K, ldc=N, batch=batch, strideA=M * K, strideW=N * K, strideC=M * N) if batch > 1 else {}
    o32 = torch.empty((batch, M, N), dtype=f32, device=dev)
    ops.gemm_x3(A if batch > 1 else A.view(A.hi[0]), Wt, o32, bias=bias.to(dev), residual=res.to(dev), res_rows=25, **kw)
    want = ref + bias.double() + res.double()[torch.arange(M) % 25]
    assert float(((o32.cpu().double() - want).abs() / (bound + 1e-3)).max()) < 2e-6
    o2 = torch.empty_like(o32)
    ops.gemm_x3(A if batch > 1 else A.view(A.hi[0]), Wt, o2, bias=bias.to(dev), residual=res.to(dev), res_rows=25, **kw)
    assert torch.equal(o32, o2)
    if (M * N) % 4 == 0:                                             # the C ABI wants the lo plane of a split output 8-byte aligned
        osp = Act.empty((batch, M, N), True, dev)
        ops.gemm_x3(A if batch > 1 else A.view(A.hi[0]), Wt, osp if batch > 1 else osp.view(osp.hi[0]), bias=bias.to(dev), act=ops.ACT_RELU, **kw)
        got = osp.t[0].float().cpu().double() + osp.t[1].float().cpu().double()
        wr = torch.relu(ref + bias.double())
        assert float(((got - wr).abs() / (bound + 1e-3 + wr.abs())).max()) < 2e-6
    o16 = Act.empty((batch, M, N), False, dev)
    ops.gemm_x3(A if batch > 1 else A.view(A.hi[0]), Wt, o16 if batch > 1 else o16.view(o16.hi[0]), bias=bias.to(dev), act=ops.ACT_QUICKGELU, **kw)
    y = ref + bias.double()
    assert torch.allclose(o16.t[0].float().cpu().double(), y * torch.sigmoid(1.702 * y), atol=4e-3, rtol=2e-3)
    if batch == 1:                                                   # fp16-valued weight: one plane (x2 form) == two planes with a zero lo plane
        Wh = W32[0].to(f16).float()
        w1, w2 = ops.split_weight(Wh.to(dev)), ops.split_weight(Wh.to(dev), allow_x2=False)
        assert w1.x2 and not w2.x2
        a, b = torch.empty((M, N), dtype=f32, device=dev), torch.empty((M, N), dtype=f32, device=dev)
        ops.gemm_x3(A.view(A.hi[0]), w1, a, bias=bias.to(dev))
        ops.gemm_x3(A.view(A.hi[0]), w2, b, bias=bias.to(dev))
        assert torch.equal(a, b)


@pytest.mark.parametrize("D,S", [(768, 1), (768, 4), (384, 2), (1024, 3), (192, 8)])
def test_sum_layernorm(dev, D, S):
    """zh_sum_layernorm_f32: planes + bias + residual -> sum (in place over the residual), LN -> fp32 / split pair with the drop-first
    row map (ln_post), chained second LN with the stacked row map (norm3 -> decoder.norm)."""
    import torch.nn.functional as F
    from zutis_amd import ops
    from zutis_amd.ops import Act
    B, T = 3, 13
    R = B * T
    parts = _randn((S, R, D), 80)
    bias, res = _randn((D,), 81), _randn((R, D), 82) * 3 + 0.5
    g1, b1, g2, b2 = _randn((D,), 83) * 0.1 + 1, _randn((D,), 84) * 0.1, _randn((D,), 85) * 0.2 + 1, _randn((D,), 86) * 0.1
    x = parts[0].clone()
    for s_ in range(1, S):
        x = x + parts[s_]
    x = (x + bias) + res                                             # the kernel's order: fp32, plane order, bias, residual
    y = F.layer_norm(x.double(), (D,), g1.double(), b1.double(), 1e-5)
    z = F.layer_norm(y, (D,), g2.double(), b2.double(), 1e-6)
    xs = res.clone().to(dev)
    o32 = torch.empty((R, D), dtype=f32, device=dev)
    o16 = Act.empty((R, D), True, dev)
    z32 = torch.zeros((B, 2, T, D), dtype=f32, device=dev)
    z16 = Act(torch.zeros((2, B * 2 * T, D), dtype=f16, device=dev))
    ops.sum_layernorm(parts.to(dev), S, R, D, bias=bias.to(dev), residual=xs, out_sum=xs, gamma=g1.to(dev), beta=b1.to(dev), eps=1e-5,
                      out_f32=o32, out_f16=o16, gamma2=g2.to(dev), beta2=b2.to(dev), eps2=1e-6, out2_f32=z32, out2_f16=z16,
                      out2_group_rows=T, out2_group_stride=2 * T, out2_offset=T)
    assert torch.equal(xs.cpu(), x)                                  # the sum is the fp32 sum in the stated order, bit for bit
    assert float((o32.cpu().double() - y).abs().max()) < 2e-5
    assert float(((o16.t[0].float() + o16.t[1].float()).cpu().double() - y).abs().max()) < 2e-5
    assert torch.equal(o16.t[0].cpu(), o32.cpu().to(f16))
    assert float((z32.cpu()[:, 1].reshape(R, D).double() - z).abs().max()) < 4e-5 and torch.all(z32[:, 0] == 0)
    zz = (z16.t[0].float() + z16.t[1].float()).cpu().view(B, 2, T, D)
    assert float((zz[:, 1].reshape(R, D).double() - z).abs().max()) < 4e-5 and torch.all(zz[:, 0] == 0)
    # ln_post's row map: the first row of every group of T (cls) is dropped, the rest packed [B, T-1]
    tok = torch.full((B * (T - 1), D), 7.0, dtype=f32, device=dev)
    ops.sum_layernorm(parts.to(dev), S, R, D, bias=bias.to(dev), residual=res.to(dev), gamma=g1.to(dev), beta=b1.to(dev), eps=1e-5, out_f32=tok,
                      out_group_rows=T, out_group_stride=T - 1, out_offset=-1, skip_first_in_group=True)
    assert float((tok.cpu().double() - y.view(B, T, D)[:, 1:].reshape(-1, D)).abs().max()) < 2e-5
    # sum only
    xo = torch.empty((R, D), dtype=f32, device=dev)
    ops.sum_layernorm(parts.to(dev), S, R, D, bias=bias.to(dev), residual=res.to(dev), out_sum=xo)
    assert torch.equal(xo.cpu(), x)


def test_split_k_planes_plus_sum_layernorm_equals_gemm_plus_layernorm(dev):
    """The few-row regime of a transformer block: the N = D GEMM as a batched GEMM over K slabs (fp32 planes) + zh_sum_layernorm_f32
    against the one-launch GEMM (bias + residual epilogue) + LayerNorm: same numbers up to fp32 re-association, fp32-class vs float64."""
    import torch.nn.functional as F
    from zutis_amd import ops
    from zutis_amd.ops import Act
    R, D, K, S = 1201, 768, 3072, 4
    A32, W32 = _randn((R, K), 90, 0.5), _randn((D, K), 91, 0.02)
    bias, X0 = _randn((D,), 92), _randn((R, D), 93)
    g, b = _randn((D,), 94) * 0.1 + 1, _randn((D,), 95) * 0.1
    A, W = _split_act(A32, dev), ops.split_weight(W32.to(dev))
    X1 = X0.clone().to(dev)
    ops.gemm_x3(A, W, X1, bias=bias.to(dev), residual=X1)
    Y1 = Act.empty((R, D), True, dev)
    ops.layernorm(X1, g.to(dev), b.to(dev), 1e-5, R, D, out_f16=Y1)
    parts = torch.empty((S, R, D), dtype=f32, device=dev)
    ops.gemm_x3(A, W, parts, M=R, N=D, K=K // S, lda=K, ldw=K, ldc=D, batch=S, strideA=K // S, strideW=K // S, strideC=R * D)
    X2 = X0.clone().to(dev)
    Y2 = Act.empty((R, D), True, dev)
    ops.sum_layernorm(parts, S, R, D, bias=bias.to(dev), residual=X2, out_sum=X2, gamma=g.to(dev), beta=b.to(dev), eps=1e-5, out_f16=Y2)
    ref = A32.double() @ W32.double().t() + bias.double() + X0.double()
    bound = float((A32.abs().double() @ W32.abs().double().t()).max())
    assert float((X2.cpu().double() - ref).abs().max()) < 2e-6 * bound + 1e-6 and float((X1.cpu().double() - ref).abs().max()) < 2e-6 * bound + 1e-6
    yr = F.layer_norm(ref, (D,), g.double(), b.double(), 1e-5)
    for Y in (Y1, Y2):
        assert float(((Y.t[0].float() + Y.t[1].float()).cpu().double() - yr).abs().max()) < 5e-6


# ----------------------------------------------------------------------------------------------- the x3 envelope (round 4)
@pytest.mark.parametrize("scale", [3e4, 1.0, 1e-4])
def test_gemm_x3_operand_magnitudes(dev, scale):
    """The split-pair GEMM at the edges of its envelope: activations of magnitude ~3e4 (just inside the fp16 range of the hi plane) stay
    fp32-class; uniformly tiny activations (1e-4: the lo halves would be subnormal fp16 numbers, ~12 bits for the pair) stay fp32-class
    when their producer stores them with a power-of-two scale (Act.out_scale, what the engine does for its unit-norm tensors) — and
    measurably do not without it."""
    from zutis_amd import ops
    from zutis_amd.ops import Act
    M, N, K = 300, 256, 512
    A32 = _randn((M, K), 1) * scale * 0.3
    W32 = _randn((N, K), 2, 0.05)
    ref = A32.double() @ W32.double().t()
    bound = A32.abs().double() @ W32.abs().double().t()
    W = ops.split_weight(W32.to(dev))

    def run(pre):                                                       # pre: the power of two the producer stores A with
        a = A32 * pre
        A = Act(torch.stack([a.to(f16), (a - a.to(f16).float()).to(f16)]).contiguous().to(dev), out_scale=1.0 / pre)
        out = torch.empty((M, N), dtype=f32, device=dev)
        ops.gemm_x3(A, W, out)
        return float(((out.cpu().double() - ref).abs() / bound).max())
    if scale >= 1.0:
        assert float(A32.abs().max()) < 65504 and run(1.0) < 2e-6
    else:
        plain, scaled = run(1.0), run(2.0 ** 13)
        assert scaled < 2e-6 and plain > 20 * scaled, (plain, scaled)


def test_unit_norm_producers_store_normal_lo_halves(dev):
    """zh_l2norm_rows / zh_global_ln_l2 / zh_cast_f32_f16 with f16_scale = 2^10 (the engine's q16 / pt16 / text16): hi + lo times 2^-10
    reproduces the fp32 value to 2^-21 relative per element (unscaled: the lo half of a ~0.04 element is subnormal, ~2^-18), and the
    class-logit product of two such pairs is fp32-class against float64."""
    from zutis_amd import ops
    from zutis_amd.ops import Act
    rows, D = 300, 512
    x = _randn((rows, D), 5)
    y32 = torch.empty((rows, D), dtype=f32, device=dev)
    errs = {}
    for name, sc in (("scaled", ops.UNIT_NORM_SCALE), ("plain", 1.0)):
        a = Act.empty((rows, D), True, dev)
        a.out_scale = 1.0 / sc
        ops.l2norm_rows(x.to(dev), rows, D, out_f32=y32, out_f16=a)
        back = (a.t[0].float() + a.t[1].float()) * a.out_scale
        errs[name] = float(((back - y32).abs() / y32.abs().clamp_min(1e-3)).max())
    assert errs["scaled"] < 2.0 ** -21 and errs["plain"] > 4 * errs["scaled"], errs
    # cast (text rows) x global-LN + L2 (patch tokens) -> logits
    t = _randn((81, D), 6); t = t / t.norm(dim=1, keepdim=True)
    B, Mpix = 1, 400
    ts = _randn((B, Mpix, D), 7) * 2 + 0.3
    t16 = Act.empty((81, D), True, dev); t16.out_scale = 1.0 / ops.UNIT_NORM_SCALE
    ops.cast_f16(t.to(dev), t16, 81, D)
    pt = torch.empty((B, Mpix, D), dtype=f32, device=dev)
    pt16 = Act.empty((B * Mpix, D), True, dev); pt16.out_scale = 1.0 / ops.UNIT_NORM_SCALE
    st = torch.zeros((1,), dtype=torch.int32, device=dev)
    ops.global_ln_l2(ts.to(dev), B, Mpix, D, out_f32=pt, out_f16=pt16, status=st)
    assert int(st.item()) == 0
    lo = torch.empty((81, Mpix), dtype=f32, device=dev)
    ops.gemm_x3(t16, pt16, lo)
    ref = t.double() @ pt.cpu()[0].double().t()
    assert float((lo.cpu().double() - ref).abs().max()) < 2e-7


def test_status_word_fires_on_non_finite_rows_only(dev):
    """The LayerNorm family raises ZH_STATUS_NONFINITE for a row holding an inf / NaN (what an fp16 split pair beyond 65504 turns into one
    product later) and leaves the word alone for finite rows of any magnitude."""
    from zutis_amd import ops
    rows, D = 9, 768
    g, b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    for val, fires in ((3e4, False), (7e4, False), (float("inf"), True), (float("nan"), True), (-float("inf"), True)):
        x = _randn((rows, D), 8)
        x[4, 100] = val                                                 # fp32 rows: 7e4 is an ordinary number HERE (the residual stream is fp32)
        st = torch.zeros((1,), dtype=torch.int32, device=dev)
        o = torch.empty((rows, D), dtype=f32, device=dev)
        ops.layernorm(x.to(dev), g, b, 1e-5, rows, D, out_f32=o, status=st)
        assert bool(int(st.item()) & ops.STATUS_NONFINITE) == fires, val
        st.zero_()
        ops.sum_layernorm(x.to(dev), 1, rows, D, gamma=g, beta=b, out_f32=o, status=st)
        assert bool(int(st.item()) & ops.STATUS_NONFINITE) == fires, val
    # ... and 7e4 in an fp16 PRODUCER is the overflow: the pair stores inf, the next GEMM returns NaN, the next LayerNorm fires
    from zutis_amd.ops import Act
    x = _randn((rows, D), 9); x[2, 5] = 7e4
    a = Act.empty((rows, D), True, dev)
    ops.cast_f16(x.to(dev), a, rows, D)
    assert torch.isinf(a.t[0][2, 5])
    out = torch.empty((rows, D), dtype=f32, device=dev)
    ops.gemm_x3(a, ops.split_weight(_randn((D, D), 10, 0.03).to(dev)), out)
    st = torch.zeros((1,), dtype=torch.int32, device=dev)
    ops.layernorm(out, g, b, 1e-5, rows, D, out_f32=torch.empty_like(out), status=st)
    assert int(st.item()) & ops.STATUS_NONFINITE


def test_engine_refuses_to_answer_outside_the_envelope(dev):
    """End to end: an MLP weight blown up so that QuickGELU(c_fc) leaves the fp16 range -> the forward's status word is set, check_finite()
    and the drop-in's predict raise (no silent garbage, no fallback); the same engine answers normally again once the weight is back."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zutis_amd", "dropin"))
    from networks.zutis import ZUTIS
    from zutis_amd import detgen, _lib
    cfg = detgen.TINY
    net = ZUTIS(categories=[f"c{i}" for i in range(5)], clip_arch="ViT-B/16", n_queries=cfg.n_queries, n_decoder_layers=cfg.dec_layers,
                n_heads=cfg.dec_heads, device=dev, text_embeddings=torch.from_numpy(detgen.text_embeddings(5, cfg.embed_dim)),
                vision_config=(cfg.width, cfg.layers, cfg.patch, cfg.grid, cfg.embed_dim))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in detgen.zutis_state_dict(cfg).items()}, strict=True)
    net = net.to(dev).eval().requires_grad_(False)
    x = torch.from_numpy(detgen.images(1, 64, 96)).to(dev)
    ok = net.predict(net(x), mask_type="semantic", size=(64, 96))
    w = net.encoder.transformer.resblocks[0].mlp.c_fc.weight
    w0 = w.detach().clone()
    with torch.no_grad():
        w.copy_(w0 * 3e5)
    for mask_type in ("semantic", "instance"):
        out = net(x)
        with pytest.raises(_lib.ZutisHipError, match="non-finite"):
            net.predict(out, mask_type=mask_type, size=(64, 96))
    net(x)
    with pytest.raises(_lib.ZutisHipError, match="non-finite"):
        net._get_engine().check_finite()
    net._get_engine().check_finite()                                    # the word was cleared by the raise
    with torch.no_grad():
        w.copy_(w0)
    again = net.predict(net(x), mask_type="semantic", size=(64, 96))
    assert np.array_equal(ok, again)
    xn = x.clone(); xn[0, 1, 3, 4] = float("nan")
    with pytest.raises(_lib.ZutisHipError, match="non-finite"):
        net.predict(net(xn), mask_type="semantic", size=(64, 96))
