"""-m gpu: every HIP kernel (through the C ABI) against a plain fp32 torch / oracle reference of the same op."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

f16, f32 = torch.float16, torch.float32


def _randn(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 192), (442, 768, 768), (100, 1764, 768), (81, 1764, 512), (37, 50, 128)])
@pytest.mark.parametrize("out_f16", [False, True])
def test_gemm_plain(dev, M, N, K, out_f16):
    from zutis_amd import ops
    A, W = _randn((M, K), 1).to(f16), _randn((N, K), 2).to(f16)
    ref = A.float() @ W.float().t()
    out = torch.empty((M, N), dtype=f16 if out_f16 else f32, device=dev)
    ops.gemm(A.to(dev), W.to(dev), out)
    tol = 2e-2 * math.sqrt(K / 64) if out_f16 else 1e-3
    assert torch.allclose(out.float().cpu(), ref, atol=tol, rtol=2e-3 if out_f16 else 1e-4)


@pytest.mark.parametrize("act", [0, 1, 2, 3, 4])
def test_gemm_epilogues(dev, act):
    """bias + activation + row-periodic residual.  Instantiated epilogues: f32 out {none, sigmoid}, f16 out
    {none, quickgelu, relu, gelu_erf} (the pairs the hot path uses); other pairs are argument errors."""
    from zutis_amd import ops, _lib
    M, N, K = 200, 260, 128
    A, W = _randn((M, K), 3, 0.5).to(f16), _randn((N, K), 4, 0.2).to(f16)
    bias, res = _randn((N,), 5), _randn((50, N), 6)
    y = A.float() @ W.float().t() + bias
    y = [y, y * torch.sigmoid(1.702 * y), F.relu(y), torch.sigmoid(y), F.gelu(y)][act]
    ref = y + res[torch.arange(M) % 50]
    for odt in (f32, f16):
        out = torch.empty((M, N), dtype=odt, device=dev)
        supported = act in ((0, 3) if odt == f32 else (0, 1, 2, 4))
        if not supported:
            with pytest.raises(_lib.ZutisHipError):
                ops.gemm(A.to(dev), W.to(dev), out, bias=bias.to(dev), residual=res.to(dev), res_rows=50, act=act)
            continue
        ops.gemm(A.to(dev), W.to(dev), out, bias=bias.to(dev), residual=res.to(dev), res_rows=50, act=act)
        if odt == f32:
            assert torch.allclose(out.cpu(), ref, atol=2e-4, rtol=1e-4)
        else:
            assert torch.allclose(out.float().cpu(), ref, atol=1e-2, rtol=2e-3)


@pytest.mark.parametrize("N,act", [(256, 0), (264, 2), (40, 0)])
def test_gemm_pos_tables(dev, N, act):
    """out[m] = act(A W^T + bias + Ty[(m % (h*w)) // w] + Tx[m % w]): the separable `pos @ Wk^T` term of the composed decoder
    K projection (engine._decoder_kv).  f32 and fp16 outputs; the fp16 one must be the single rounding of the fp32 one."""
    from zutis_amd import ops
    hh, ww, B, K = 6, 10, 5, 128
    M = B * hh * ww - 7                                                     # last image ragged: rows clamp, not wrap
    A, W = _randn((M, K), 31, 0.5).to(f16), _randn((N, K), 32, 0.2).to(f16)
    bias, Ty, Tx = _randn((N,), 33), _randn((hh, N), 34), _randn((ww, N), 35)
    m = torch.arange(M)
    ref = A.double() @ W.double().t() + bias.double() + Ty.double()[(m % (hh * ww)) // ww] + Tx.double()[m % ww]
    ref = F.relu(ref) if act == 2 else ref
    kw = dict(bias=bias.to(dev), pos=(Ty.to(dev), Tx.to(dev)), act=act)
    o16 = torch.empty((M, N), dtype=f16, device=dev)
    ops.gemm(A.to(dev), W.to(dev), o16, **kw)
    assert float((o16.cpu().double() - ref).abs().max()) < 1e-2
    if act == 0:
        o32 = torch.empty((M, N), dtype=f32, device=dev)
        ops.gemm(A.to(dev), W.to(dev), o32, **kw)
        assert float((o32.cpu().double() - ref).abs().max()) < 2e-4
        assert torch.equal(o16.cpu(), o32.cpu().to(f16))
    # fp16 tables (what the engine uses when K itself is fp16): same sum from the rounded tables
    Th, Tw = Ty.to(f16), Tx.to(f16)
    ref16 = A.double() @ W.double().t() + bias.double() + Th.double()[(m % (hh * ww)) // ww] + Tw.double()[m % ww]
    ref16 = F.relu(ref16) if act == 2 else ref16
    o32 = torch.empty((M, N), dtype=f32, device=dev)
    if act == 0:
        ops.gemm(A.to(dev), W.to(dev), o32, bias=bias.to(dev), pos=(Th.to(dev), Tw.to(dev)))
        assert float((o32.cpu().double() - ref16).abs().max()) < 2e-4


def test_gemm_inplace_residual_and_batched(dev):
    from zutis_amd import ops
    B, M, N, K = 3, 100, 140, 64
    A, W = _randn((B, M, K), 7).to(f16), _randn((B, N, K), 8).to(f16)
    X = _randn((B, M, N), 9)
    ref = torch.einsum("bmk,bnk->bmn", A.float(), W.float()) + X
    Xd = X.to(dev).contiguous()
    ops.gemm(A.to(dev), W.to(dev), Xd, residual=Xd, res_rows=M, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, ldr=N, batch=B,
             strideA=M * K, strideW=N * K, strideC=M * N, strideR=M * N)
    assert torch.allclose(Xd.cpu(), ref, atol=1e-3, rtol=1e-4)
    # shared A (stride 0), unaligned N -> scalar epilogue
    N2 = 37
    W2 = _randn((B, N2, K), 10).to(f16)
    out = torch.empty((B, M, N2), dtype=f32, device=dev)
    ops.gemm(A[0].to(dev), W2.to(dev), out, M=M, N=N2, K=K, lda=K, ldw=K, ldc=N2, batch=B, strideA=0, strideW=N2 * K, strideC=M * N2)
    assert torch.allclose(out.cpu(), torch.einsum("mk,bnk->bmn", A[0].float(), W2.float()), atol=1e-3, rtol=1e-4)


def test_gemm_race_screen_bitwise_repeatable(dev):
    """The K loop is an LDS-DMA ring retired by counted vmcnt + raw barriers: a mis-counted wait shows up as RARE wrong
    tiles.  Screen every tile shape (256x256, 256x192, 128x128, 128x64) over many launches: results must be bitwise
    identical run to run and match the fp32 reference."""
    from zutis_amd import ops
    for (M, N, K) in [(2048, 2304, 768), (1500, 768, 3072), (700, 640, 192), (300, 200, 64), (3200, 768, 128)]:
        A, W = _randn((M, K), 100 + M).to(f16).to(dev), _randn((N, K), 200 + N).to(f16).to(dev)
        ref = (A.float() @ W.float().t())
        first = None
        for it in range(40):
            out = torch.empty((M, N), dtype=f32, device=dev)
            ops.gemm(A, W, out)
            if first is None:
                first = out.clone()
                assert torch.allclose(first, ref, atol=2e-3 * math.sqrt(K / 64), rtol=1e-3)
            else:
                assert torch.equal(out, first), f"non-repeatable GEMM result at launch {it} for {(M, N, K)}"


@pytest.mark.parametrize("tile", ["256", "192", "128", "64", "2128", "2064", "3064", "7032", "7096", "7128"])
def test_gemm_every_tile_variant_every_ring_phase(dev, tile):
    """Each tile / ring-depth variant (forced through the developer entry zh_dev_set_gemm_overrides: the environment is read
    once per process) over K = 64 .. 1024: every prologue / steady / tail combination of the 4- and 8-deep LDS-DMA rings and of the
    64-k-slice rings of round 5 (7032 / 7096 / 7128: five and seven slots), ragged M and N, repeated launches bitwise identical — and
    bitwise the 128 x 128 tile's result (the K order inside a tile does not depend on the tile: what the tail peel and the
    batch-invariance of the engine rest on)."""
    from zutis_amd import ops, _lib
    L = _lib.load(raw=True)
    try:
        M, N = 333, 328
        for K in (64, 128, 192, 256, 320, 448, 512, 576, 640, 1024):
            A, W = _randn((M, K), 300 + K, 0.5).to(f16).to(dev), _randn((N, K), 400 + K, 0.5).to(f16).to(dev)
            ref = A.float() @ W.float().t()
            _lib.check(L.zh_dev_set_gemm_overrides(0, 128, 0), "zh_dev_set_gemm_overrides")
            base = torch.empty((M, N), dtype=f32, device=dev)
            ops.gemm(A, W, base)
            _lib.check(L.zh_dev_set_gemm_overrides(0, int(tile), 0), "zh_dev_set_gemm_overrides")
            outs = []
            for _ in range(3):
                out = torch.empty((M, N), dtype=f32, device=dev)
                ops.gemm(A, W, out)
                outs.append(out)
            assert torch.allclose(outs[0], ref, atol=1e-3 * math.sqrt(K / 64), rtol=1e-3), (tile, K)
            assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (tile, K)
            assert torch.equal(outs[0], base), (tile, K)
    finally:
        L.zh_dev_set_gemm_overrides(0, 0, 0)


def test_gemm_tail_peel_is_bitwise_one_launch(dev):
    """zh_gemm_f16's tail peel (round 5; the f16x3 dispatcher has had it since round 3): 257 m-tiles x 4 n-tiles of 256 x 256 = 4 rounds
    of the chip + 4 tiles run as the whole rounds + a second call on the last m-tile row — bitwise the single forced-tile launch, for
    the fp32 + residual and the fp16 + activation epilogues, ragged last tile included."""
    from zutis_amd import ops, _lib
    L = _lib.load(raw=True)
    M, N, K = 257 * 256 - 37, 1024, 128
    A, W = _randn((M, K), 1, 0.5).to(f16).to(dev), _randn((N, K), 2, 0.05).to(f16).to(dev)
    bias, res = _randn((N,), 3).to(dev), _randn((M, N), 4).to(dev)

    def run():
        o = torch.empty((M, N), dtype=f32, device=dev)
        ops.gemm(A, W, o, bias=bias, residual=res)
        h = torch.empty((M, N), dtype=f16, device=dev)
        ops.gemm(A, W, h, bias=bias, act=ops.ACT_QUICKGELU)
        return o, h
    o_p, h_p = run()                                    # cost model: 256 x 256 tiles, 1028 of them -> peeled
    try:
        _lib.check(L.zh_dev_set_gemm_overrides(0, 256, 0), "zh_dev_set_gemm_overrides")
        o_1, h_1 = run()                                # forced tile: one launch, no peel
    finally:
        L.zh_dev_set_gemm_overrides(0, 0, 0)
    assert torch.equal(o_p, o_1) and torch.equal(h_p, h_1)
    ref = A.double() @ W.double().t() + bias.double() + res.double()
    assert float((o_p.double() - ref).abs().max()) < 1e-3


@pytest.mark.parametrize("tile,N", [("256", 1024), ("192", 960)])
def test_gemm_persistent_tiles_are_bitwise_one_workgroup_per_tile(dev, tile, N):
    """The big plain-fp16 tiles walk a launch's tiles with persistent workgroups and request the next tile's first K slices under the
    current tile's epilogue (gemm_kernel.h PERS).  Forced grids of 8 / 16 / 40 workgroups (3 - 20 tiles each, ragged last tiles, K from
    one slice to many, batched) against one workgroup per tile (persist = 0): bit for bit, for the fp16 + activation epilogue and — on
    the 256 x 192 tile — fp32 + bias + residual; repeated launches identical (the LDS hand-over between epilogue slabs and the next
    prologue is what a race would break)."""
    from zutis_amd import ops, _lib
    L = _lib.load(raw=True)
    M = 5 * 256 + 77
    try:
        _lib.check(L.zh_dev_set_gemm_overrides(0, int(tile), 0), "zh_dev_set_gemm_overrides")
        for K, batch in ((64, 1), (128, 1), (192, 2), (768, 1), (1024, 3)):
            A = _randn((batch, M, K), 500 + K, 0.5).to(f16).to(dev)
            W = _randn((batch, N, K), 600 + K, 0.05).to(f16).to(dev)
            bias = _randn((N,), 7).to(dev)
            res = _randn((M, N), 8).to(dev)

            def run():
                h = torch.empty((batch, M, N), dtype=f16, device=dev)
                ops.gemm(A, W, h, bias=bias, act=ops.ACT_QUICKGELU, batch=batch, strideA=M * K, strideW=N * K, strideC=M * N)
                o = torch.empty((batch, M, N), dtype=f32, device=dev)
                ops.gemm(A, W, o, bias=bias, residual=res, res_rows=M, batch=batch, strideA=M * K, strideW=N * K, strideC=M * N)
                return h, o
            _lib.check(L.zh_dev_set_gemm_persist(0), "zh_dev_set_gemm_persist")
            h0, o0 = run()
            ref = torch.einsum("bmk,bnk->bmn", A.float(), W.float()) + bias + res
            assert torch.allclose(o0, ref, atol=2e-3 * math.sqrt(K / 64), rtol=1e-3)
            for g in (8, 16, 40):
                _lib.check(L.zh_dev_set_gemm_persist(g), "zh_dev_set_gemm_persist")
                for _ in range(3):
                    h1, o1 = run()
                    assert torch.equal(h1, h0) and torch.equal(o1, o0), (tile, K, batch, g)
    finally:
        L.zh_dev_set_gemm_overrides(0, 0, 0)
        L.zh_dev_set_gemm_persist(-1)            # back to the default: the device's CU count


def test_gemm_forced_tile_is_really_forced(dev):
    """The override reaches the launcher: an unknown tile code is an argument error (it would be ignored if the setter did
    nothing), and clearing it restores the cost model."""
    from zutis_amd import ops, _lib
    L = _lib.load(raw=True)
    A, W = torch.zeros((64, 64), dtype=f16, device=dev), torch.zeros((64, 64), dtype=f16, device=dev)
    out = torch.empty((64, 64), dtype=f32, device=dev)
    L.zh_dev_set_gemm_overrides(0, 777, 0)
    try:
        with pytest.raises(_lib.ZutisHipError, match="not a tile code"):
            ops.gemm(A, W, out)
    finally:
        L.zh_dev_set_gemm_overrides(0, 0, 0)
    ops.gemm(A, W, out)


def test_gemm_rejects_bad_k(dev):
    from zutis_amd import ops, _lib
    A, W = torch.zeros((8, 40), dtype=f16, device=dev), torch.zeros((8, 40), dtype=f16, device=dev)
    with pytest.raises(_lib.ZutisHipError):
        ops.gemm(A, W, torch.empty((8, 8), dtype=f32, device=dev))


@pytest.mark.parametrize("dh,H,Tq,Tk", [(64, 3, 50, 50), (64, 12, 442, 442), (96, 2, 5, 140), (96, 8, 100, 1764), (96, 8, 100, 100), (64, 2, 130, 67)])
def test_attention(dev, dh, H, Tq, Tk):
    from zutis_amd import ops
    B, D = 2, H * dh
    q, k, v = _randn((B, Tq, D), 11).to(f16), _randn((B, Tk, D), 12).to(f16), _randn((B, Tk, D), 13).to(f16)
    qh, kh, vh = (t.float().view(B, -1, H, dh).transpose(1, 2) for t in (q, k, v))
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(dh), -1) @ vh).transpose(1, 2).reshape(B, Tq, D)
    o = torch.empty((B, Tq, D), dtype=f16, device=dev)
    ops.attention(q.to(dev), k.to(dev), v.to(dev), o, batch=B, heads=H, Tq=Tq, Tk=Tk, head_dim=dh, ldq=D, ldk=D, ldv=D, ldo=D,
                  strideQ=Tq * D, strideK=Tk * D, strideV=Tk * D, strideO=Tq * D)
    err = (o.float().cpu() - ref).abs().max().item()
    assert err < 4e-3, err


def test_attention_spike_forces_rescale(dev):
    """One key far above the rest in a LATE tile forces the online-softmax rescale branch (and alpha != 1)."""
    from zutis_amd import ops
    B, H, dh, Tq, Tk = 1, 1, 64, 32, 300
    q, k, v = _randn((B, Tq, dh), 21), _randn((B, Tk, dh), 22) * 0.1, _randn((B, Tk, dh), 23)
    k[0, 250] = q[0, 3] * 4.0
    q, k, v = q.to(f16), k.to(f16), v.to(f16)
    ref = torch.softmax(q.float() @ k.float().transpose(-1, -2) / 8.0, -1) @ v.float()
    o = torch.empty((B, Tq, dh), dtype=f16, device=dev)
    ops.attention(q.to(dev), k.to(dev), v.to(dev), o, batch=B, heads=H, Tq=Tq, Tk=Tk, head_dim=dh, ldq=dh, ldk=dh, ldv=dh, ldo=dh,
                  strideQ=Tq * dh, strideK=Tk * dh, strideV=Tk * dh, strideO=Tq * dh)
    assert (o.float().cpu() - ref).abs().max().item() < 4e-3


@pytest.mark.parametrize("D", [192, 384, 768, 1024])
def test_layernorm(dev, D):
    from zutis_amd import ops
    B, T = 3, 11
    x, g, b = _randn((B * T, D), 31) * 3 + 1, _randn((D,), 32) * 0.1 + 1, _randn((D,), 33) * 0.1
    add = _randn((T, D), 34)
    ref = F.layer_norm(x, (D,), g, b, 1e-5)
    o32 = torch.empty((B * T, D), dtype=f32, device=dev)
    o16 = torch.empty((B * T, D), dtype=f16, device=dev)
    p16 = torch.empty((B * T, D), dtype=f16, device=dev)
    p32 = torch.empty((B * T, D), dtype=f32, device=dev)
    ops.layernorm(x.to(dev), g.to(dev), b.to(dev), 1e-5, B * T, D, out_f32=o32, out_f16=o16, out_f16_plus=p16, out_f32_plus=p32,
                  add=add.to(dev), add_rows=T)
    assert torch.allclose(o32.cpu(), ref, atol=2e-5, rtol=1e-5)
    assert torch.allclose(o16.float().cpu(), ref, atol=4e-3, rtol=2e-3)
    assert torch.allclose(p32.cpu(), ref + add.repeat(B, 1), atol=2e-5, rtol=1e-5)
    assert torch.allclose(p16.float().cpu(), ref + add.repeat(B, 1), atol=6e-3, rtol=2e-3)
    # drop-cls input mapping + stacked output mapping, no affine, eps 1e-6
    o = torch.zeros((B * 2 * (T - 1), D), dtype=f32, device=dev)
    ops.layernorm(x.to(dev), None, None, 1e-6, B * (T - 1), D, out_f32=o, in_group_rows=T - 1, in_group_stride=T, in_offset=1,
                  out_group_rows=T - 1, out_group_stride=2 * (T - 1), out_offset=T - 1)
    ref2 = F.layer_norm(x.view(B, T, D)[:, 1:], (D,), None, None, 1e-6)
    got = o.cpu().view(B, 2, T - 1, D)
    assert torch.allclose(got[:, 1], ref2, atol=2e-5, rtol=1e-5)
    assert torch.all(got[:, 0] == 0)


def test_assemble_tokens_ln(dev):
    from zutis_amd import ops
    B, hw, D = 2, 35, 192
    pe, cls, pos = _randn((B * hw, D), 41), _randn((D,), 42), _randn((1 + hw, D), 43)
    g, b = _randn((D,), 44) * 0.1 + 1, _randn((D,), 45) * 0.1
    t = torch.cat([cls[None, None].expand(B, 1, D), pe.view(B, hw, D)], 1) + pos[None]
    ref = F.layer_norm(t, (D,), g, b, 1e-5)
    out = torch.empty((B, 1 + hw, D), dtype=f32, device=dev)
    ops.assemble_tokens_ln(pe.to(dev), cls.to(dev), pos.to(dev), g.to(dev), b.to(dev), 1e-5, out, B, 1 + hw, D)
    assert torch.allclose(out.cpu(), ref, atol=2e-5, rtol=1e-5)


def test_l2norm_and_global_ln(dev):
    from zutis_amd import ops
    x = _randn((50, 768), 51)
    o = torch.empty((50, 768), dtype=f32, device=dev)
    ops.l2norm_rows(x.to(dev), 50, 768, out_f32=o)
    assert torch.allclose(o.cpu(), x / x.norm(dim=-1, keepdim=True), atol=1e-6, rtol=1e-5)
    for (B, h, w, C) in [(2, 10, 14, 64), (3, 42, 42, 512)]:
        t = _randn((B, h, w, C), 52) * 2 + 0.3
        ref = F.layer_norm(t, t.shape[1:])
        ref = ref / (ref.norm(dim=-1, keepdim=True) + 1e-7)
        o = torch.empty((B, h, w, C), dtype=f32, device=dev)
        o16 = torch.empty((B, h, w, C), dtype=f16, device=dev)
        ops.global_ln_l2(t.to(dev), B, h * w, C, out_f32=o, out_f16=o16)
        assert torch.allclose(o.cpu(), ref, atol=2e-6, rtol=1e-5)
        assert torch.allclose(o16.float().cpu(), ref, atol=1e-3)


def test_im2col_matches_conv(dev):
    from zutis_amd import ops
    B, p, D = 2, 16, 64
    x, wc = _randn((B, 3, 80, 117), 61), _randn((D, 3, p, p), 62) * 0.05
    ref = F.conv2d(x.half().float(), wc.half().float(), stride=p)
    gh, gw = ref.shape[-2:]
    col = torch.empty((B * gh * gw, 3 * p * p), dtype=f16, device=dev)
    ops.im2col(x.to(dev), col, p, 3 * p * p)
    out = torch.empty((B * gh * gw, D), dtype=f32, device=dev)
    ops.gemm(col, wc.reshape(D, -1).to(f16).to(dev), out)
    assert torch.allclose(out.cpu().view(B, gh * gw, D), ref.flatten(2).transpose(1, 2), atol=2e-3, rtol=1e-3)


def test_posembed_bicubic_vs_golden_and_oracle(dev, golden_dir):
    from zutis_amd import ops, detgen
    from oracle import resample as R
    g = np.load(f"{golden_dir}/ops.npz")
    for gr, (h, w) in [(14, (21, 21)), (14, (32, 32)), (7, (7, 7)), (14, (30, 40)), (4, (5, 7))]:
        pe = detgen.det_normal(f"pe_{gr}", (gr * gr + 1, 48))
        out = torch.empty((1 + h * w, 48), dtype=f32, device=dev)
        ops.posembed_bicubic(torch.from_numpy(pe).to(dev), out, gr, h, w, 48, np.float32(1.0 / ((h + 0.1) / gr)),
                             np.float32(1.0 / ((w + 0.1) / gr)))
        assert np.abs(out.cpu().numpy() - g[f"posembed_g{gr}_{h}x{w}"]).max() < 5e-6
        orc = R.bicubic_cl(pe[1:].reshape(gr, gr, 48), h, w, (h + 0.1) / gr, (w + 0.1) / gr).reshape(h * w, 48)
        assert np.abs(out.cpu().numpy()[1:] - orc).max() < 5e-6


def test_upsample2x_and_sine(dev, golden_dir):
    from zutis_amd import ops, detgen
    g = np.load(f"{golden_dir}/ops.npz")
    x = detgen.det_normal("up2", (2, 5, 7, 24))
    o = torch.empty((2, 10, 14, 24), dtype=f32, device=dev)
    o16 = torch.empty((2, 10, 14, 24), dtype=f16, device=dev)
    ops.upsample2x_cl(torch.from_numpy(x).to(dev), 2, 5, 7, 24, out_f32=o, out_f16=o16)
    assert np.abs(o.cpu().numpy() - g["up2"]).max() < 1e-6
    assert np.abs(o16.float().cpu().numpy() - g["up2"]).max() < 4e-3
    for (h, w, D) in [(10, 14, 96), (12, 17, 768)]:
        pe = torch.empty((h * w, D), dtype=f32, device=dev)
        ops.sine_pe(pe, h, w, D)
        assert np.abs(pe.cpu().numpy() - g[f"sine_{h}x{w}"]).max() < 2e-5


def test_upsample_argmax_bit_exact(dev, golden_dir):
    """Integer output: bit-exact against the reference's torch.argmax(F.interpolate(...)) incl. exact ties."""
    from zutis_amd import ops
    from oracle import resample as R
    g = np.load(f"{golden_dir}/ops.npz")
    lo = torch.from_numpy(g["argmax_lo"]).to(dev)
    for (H, W) in [(80, 112), (77, 145)]:
        lab = torch.empty((2, H, W), dtype=torch.int64, device=dev)
        ops.upsample_argmax(lo, lab, 2, 9, 10, 14, H, W)
        assert np.array_equal(lab.cpu().numpy(), g[f"argmax_labels_{H}x{W}"])
        full = torch.empty((2, 9, H, W), dtype=f32, device=dev)
        m = torch.empty((2, 9, H, W), dtype=torch.uint8, device=dev)
        ops.upsample_bilinear_nchw(lo, 18, 10, 14, H, W, out=full, mask_u8=m, threshold=0.5)
        ref = R.bilinear_nchw(g["argmax_lo"], H, W)
        assert np.array_equal(full.cpu().numpy(), ref)          # fp32 bilinear is bit-identical too
        assert np.array_equal(m.cpu().numpy().astype(bool), ref > 0.5)
    # identity size and a big random case against the oracle
    x = _randn((3, 81, 21, 21), 71).numpy()
    for (H, W) in [(21, 21), (336, 336), (427, 640)]:
        lab = torch.empty((3, H, W), dtype=torch.int64, device=dev)
        ops.upsample_argmax(torch.from_numpy(x).to(dev), lab, 3, 81, 21, 21, H, W)
        assert np.array_equal(lab.cpu().numpy(), R.bilinear_argmax_nchw(x, H, W))
    # many classes (class chunks with a remainder), 16x upsampling (C4 geometry: 32x32 -> 518), ragged tiles
    x = _randn((1, 150, 32, 32), 72).numpy()
    x[0, 7] = x[0, 140]                                          # exact ties across chunks: the first index must win
    lab = torch.empty((1, 518, 518), dtype=torch.int64, device=dev)
    ops.upsample_argmax(torch.from_numpy(x).to(dev), lab, 1, 150, 32, 32, 518, 518)
    ref = R.bilinear_argmax_nchw(x, 518, 518)
    assert np.array_equal(lab.cpu().numpy(), ref) and not (ref == 140).any()
    # ties INSIDE a group of four classes (the round-6 kernel resolves the index from the group's maximum: the first equal value), across
    # groups of one chunk, and between -0.0 and +0.0 (equal: the first index wins); class counts that are not multiples of four
    for n in (6, 37, 81):
        x = -np.abs(_randn((2, n, 12, 12), 74 + n).numpy()) - 0.5
        x[0, 1] = x[0, 3] = np.abs(x[0, 3])                      # same group, first and third member
        x[1, 2] = -0.0; x[1, 5] = 0.0                            # -0.0 (class 2) before +0.0 (class 5): equal, class 2 wins
        if n > 36:
            x[0, 33] = x[0, 1]                                   # and again in the next chunk: never replaces
        lab = torch.empty((2, 96, 100), dtype=torch.int64, device=dev)
        ops.upsample_argmax(torch.from_numpy(x).to(dev), lab, 2, n, 12, 12, 96, 100)
        ref = R.bilinear_argmax_nchw(x, 96, 100)
        assert np.array_equal(lab.cpu().numpy(), ref) and (ref[0] == 1).all() and (ref[1] == 2).all()
    # NaN logits: torch.argmax / np.argmax take the FIRST NaN as the maximum; -inf everywhere -> index 0
    x = _randn((2, 40, 6, 6), 73).numpy()
    x[0, 35, 2, 3] = np.nan; x[0, 3, 2, 3] = np.nan; x[1, :, 4:, 4:] = -np.inf; x[1, 20, 0, 0] = np.inf
    lab = torch.empty((2, 96, 96), dtype=torch.int64, device=dev)
    ops.upsample_argmax(torch.from_numpy(x).to(dev), lab, 2, 40, 6, 6, 96, 96)
    with np.errstate(invalid="ignore"):
        ref = R.bilinear_argmax_nchw(x, 96, 96)
    assert np.array_equal(lab.cpu().numpy(), ref) and (ref[0] == 3).any() and (ref[1][-8:, -8:] == 0).all()


def test_confusion_hist(dev, golden_dir):
    from zutis_amd import ops
    g = np.load(f"{golden_dir}/ops.npz")
    hist = torch.zeros((49,), dtype=torch.int64, device=dev)
    ops.confusion_hist(torch.from_numpy(g["rs_gt"]).to(dev).contiguous(), torch.from_numpy(g["rs_pred"]).to(dev).contiguous(), hist, 7)
    assert np.array_equal(hist.cpu().numpy().reshape(7, 7), g["rs_hist"].astype(np.int64))
    # large class count -> global-atomic path
    rng = np.random.default_rng(0)
    gt, pr = rng.integers(-1, 921, (200000,)), rng.integers(0, 920, (200000,))
    hist = torch.zeros((920 * 920,), dtype=torch.int64, device=dev)
    ops.confusion_hist(torch.from_numpy(gt).to(dev), torch.from_numpy(pr).to(dev), hist, 920)
    m = (gt >= 0) & (gt < 920)
    assert np.array_equal(hist.cpu().numpy(), np.bincount(920 * gt[m] + pr[m], minlength=920 * 920))


def test_topk_rows_exact(dev):
    """Integer output: the k largest per row, score descending / index ascending, incl. heavy ties, negatives, k == N."""
    from zutis_amd import ops
    rng = np.random.default_rng(0)
    for (R, N, k) in [(5, 1000, 500), (3, 70000, 500), (4, 513, 513), (2, 2000, 1), (3, 5000, 1000)]:
        x = rng.standard_normal((R, N)).astype(np.float32)
        x[0, : N // 2] = 0.25                                   # massive exact ties
        x[-1] = -np.abs(x[-1])                                  # all negative
        x[1 % R, 7] = np.inf
        idx, val = ops.topk_rows(torch.from_numpy(x).to(dev), k, with_values=True)
        ref = np.stack([np.lexsort((np.arange(N), -row))[:k] for row in x])
        assert np.array_equal(idx.cpu().numpy(), ref)
        assert np.array_equal(val.cpu().numpy(), np.take_along_axis(x, ref, 1))
    # padded leading dimension: only the first N columns count
    x = rng.standard_normal((2, 104)).astype(np.float32); x[:, 100:] = 1e9
    idx = ops.topk_rows(torch.from_numpy(x).to(dev), 10, N=100)
    assert np.array_equal(idx.cpu().numpy(), np.stack([np.lexsort((np.arange(100), -r[:100]))[:10] for r in x]))
    # index map + offset + strided outputs (candidate-table forms used by the retrieval merge)
    xm = torch.from_numpy(x).to(dev)
    imap = torch.arange(208, dtype=torch.int64, device=dev).view(2, 104) * 3 + 1
    wide_i = torch.full((2, 30), -7, dtype=torch.int64, device=dev); wide_v = torch.zeros((2, 30), device=dev)
    ops.topk_rows(xm, 10, N=100, with_values=True, idx_map=imap, out_idx=wide_i[:, 10:20], out_val=wide_v[:, 10:20])
    ref = np.stack([np.lexsort((np.arange(100), -r[:100]))[:10] for r in x])
    assert np.array_equal(wide_i[:, 10:20].cpu().numpy(), np.take_along_axis(imap.cpu().numpy(), ref, 1))
    assert np.array_equal(wide_v[:, 10:20].cpu().numpy(), np.take_along_axis(x, ref, 1))
    assert int((wide_i[:, :10] != -7).sum()) == 0 and int((wide_i[:, 20:] != -7).sum()) == 0
    assert np.array_equal(ops.topk_rows(xm, 10, N=100, idx_add=1000).cpu().numpy(), ref + 1000)


def test_retrieval_vs_oracle(dev):
    """text @ image.T + top-k (datasets/index_dataset.py:163-167) on UN-rounded fp32 operands against the fp32 oracle: the
    similarity GEMM runs in the reference-equivalent x3 mode, so the selected indices equal the fp32 product's wherever two
    scores are further apart than fp32 summation-order noise.  Chunked path included."""
    from zutis_amd import retrieval, detgen
    from oracle import zutis_ref as O
    C, N, E, k = 9, 20011, 768, 500
    t = detgen.text_embeddings(C, E)
    im = detgen.det_normal("imgemb", (N, E)); im /= np.linalg.norm(im, axis=1, keepdims=True)
    im[777] = im[123]                                                    # an exact tie: ascending index
    ref_idx, ref_val = O.retrieve_topk(t, im, k)                         # fp32 product
    sim64 = t.astype(np.float64) @ im.astype(np.float64).T
    for chunk in (1 << 20, 4096):
        idx, val = retrieval.retrieve_topk(torch.from_numpy(t).to(dev), torch.from_numpy(im).to(dev), k, chunk=chunk)
        got, gv = idx.cpu().numpy(), val.cpu().numpy()
        assert np.abs(gv - np.take_along_axis(sim64, got, 1)).max() < 3e-7   # fp32-class scores
        for c in range(C):
            assert list(got[c]).index(123) + 1 == list(got[c]).index(777) if 123 in got[c] and 777 in got[c] else True
            diff = np.nonzero(got[c] != ref_idx[c])[0]
            # any disagreement with the fp32 oracle must be a swap between scores closer than summation-order noise
            assert np.all(np.abs(sim64[c, got[c][diff]] - sim64[c, ref_idx[c][diff]]) < 2e-7)
            assert len(diff) <= 4
    # k larger than the image count is clamped
    idx, _ = retrieval.retrieve_topk(torch.from_numpy(t).to(dev), torch.from_numpy(im[:40]).to(dev), 500)
    assert idx.shape == (C, 40)


def test_mask_runs_device_rle_matches_host_encoder(dev):
    """Device run extraction + box + area == host RLE encoder / numpy box for random, empty, full and striped masks."""
    from zutis_amd import ops, rle
    from zutis_amd.engine import ZutisEngine
    rng = np.random.default_rng(1)
    H, W = 77, 145
    masks = (rng.random((7, H, W)) > 0.6).astype(np.uint8)
    masks[1] = 0; masks[2] = 1; masks[3] = 0; masks[3][10:30, 64:66] = 1; masks[4] = 0; masks[4][0, 0] = 1; masks[5][:, ::2] = 1
    dm = torch.from_numpy(masks).to(dev)
    sel = np.array([6, 0, 1, 2, 3, 4, 5], np.int32)
    rles, boxes, areas = ZutisEngine.encode_masks(None, dm, sel, max_runs=20000)
    for j, q in enumerate(sel):
        m = masks[q]
        assert areas[j] == int(m.sum())
        assert rles[j] == rle.encode(m)
        if m.any():
            assert boxes[j] == rle.mask_to_box(m)
    rles2, _, _ = ZutisEngine.encode_masks(None, dm, np.array([0], np.int32), max_runs=16)     # overflow -> host fallback
    assert rles2[0] == rle.encode(masks[0])


@pytest.mark.parametrize("H,W", [(480, 640), (427, 640), (700, 96), (3, 1000), (333, 64)])
def test_mask_runs_panel_blocks_at_evaluation_sizes(dev, H, W):
    """zh_mask_runs runs one block per (mask, 64-column panel): blob masks at the COCO-20K evaluation sizes (ten panels, the
    16-byte load path), a tall narrow, a 3-row and a one-panel mask: RLE strings, boxes and areas equal the host encoder's;
    masks whose runs straddle every panel boundary (horizontal stripes) and empty / full masks included."""
    from zutis_amd import rle
    from zutis_amd.engine import ZutisEngine
    rng = np.random.default_rng(H * 1000 + W)
    yy, xx = np.mgrid[:H, :W]
    masks = []
    for i in range(6):                                     # blobs: a few hundred runs, like thresholded mask proposals
        cy, cx, r = rng.uniform(0, H), rng.uniform(0, W), rng.uniform(0.1, 0.5) * max(H, W)
        masks.append((((yy - cy) ** 2 + (xx - cx) ** 2) < r ** 2) & (rng.random((H, W)) > 0.02))
    stripes = np.zeros((H, W), bool); stripes[::3] = True
    masks += [stripes, np.zeros((H, W), bool), np.ones((H, W), bool), rng.random((H, W)) > 0.5]
    m8 = np.stack(masks).astype(np.uint8)
    dm = torch.from_numpy(m8).to(dev)
    sel = np.arange(len(masks), dtype=np.int32)[::-1].copy()
    rles, boxes, areas = ZutisEngine.encode_masks(None, dm, sel, max_runs=H * W + 1)
    for j, q in enumerate(sel):
        assert areas[j] == int(m8[q].sum()), (q, areas[j])
        assert rles[j] == rle.encode(m8[q]), q
        if m8[q].any():
            assert boxes[j] == rle.mask_to_box(m8[q]), q


def test_retrieval_shard_merge_equals_unsharded(dev):
    """Per-shard device top-k + merge_topk (what retrieve_topk_sharded does after its all-gather) == retrieve_topk over all
    images, including an exact cross-shard score tie (smaller global index first) and a shard shorter than k."""
    from zutis_amd import retrieval, detgen
    C, E, N, k = 5, 64, 700, 40
    t = detgen.text_embeddings(C, E, seed=3)
    im = detgen.text_embeddings(N, E, seed=4)
    im[650] = im[10]
    td, imd = torch.from_numpy(t).to(dev), torch.from_numpy(im).to(dev)
    ref_i, ref_v = retrieval.retrieve_topk(td, imd, k)
    cand_i, cand_v = [], []
    for lo, hi in ((0, 30), (30, 400), (400, 700)):                   # first shard shorter than k
        i, v = retrieval.retrieve_topk(td, imd[lo:hi], min(k, hi - lo), index_offset=lo)
        pi = torch.full((C, k), -1, dtype=torch.int64, device=dev); pv = torch.full((C, k), float("-inf"), device=dev)
        pi[:, : i.shape[1]] = i; pv[:, : v.shape[1]] = v
        cand_i.append(pi); cand_v.append(pv)
    mi, mv = retrieval.merge_topk(torch.cat(cand_i, 1), torch.cat(cand_v, 1), k)
    assert torch.equal(mi, ref_i) and torch.equal(mv, ref_v)


def test_device_rle_matches_hand_derived_vectors(dev, golden_dir):
    """Device run extraction (zh_mask_runs) + string packing reproduce the hand-derived COCO RLE vectors."""
    import json
    from zutis_amd.engine import ZutisEngine
    vecs = json.load(open(f"{golden_dir}/rle_vectors.json"))["vectors"]
    for v in vecs:
        h, w = v["size"]
        flat = np.zeros(h * w, np.uint8)
        pos, val = 0, 0
        for r in v["runs_colmajor"]:
            flat[pos:pos + r] = val; pos += r; val ^= 1
        m = flat.reshape((h, w), order="F")
        rles, boxes, areas = ZutisEngine.encode_masks(None, torch.from_numpy(np.ascontiguousarray(m[None])).to(dev), np.array([0], np.int32))
        assert rles[0]["counts"] == v["counts"].encode("ascii") and rles[0]["size"] == v["size"], v["name"]
        assert areas[0] == int(m.sum())


@pytest.mark.parametrize("Q", [100, 150])        # <= 128 candidates: the one-wave kernel; more: the block-wide one
@pytest.mark.parametrize("nms_type", ["hard", "linear", "gaussian"])
def test_device_mask_nms_matches_reference_control_flow(dev, nms_type, Q):
    """zh_mask_nms (one workgroup per image) against the oracle's restatement of ZUTIS.non_maximum_suppression
    (zutis.py:211-299) on random overlapping masks: same (category, query) emission order; scores equal (hard: exactly —
    only x1 / x0 products; linear / gaussian: float64 arithmetic, 1e-12)."""
    from zutis_amd import ops
    from zutis_amd.engine import ZutisEngine
    from oracle import zutis_ref as O
    rng = np.random.default_rng(5)
    B, H, W = 3, 48, 64
    masks = np.zeros((B, Q, H, W), np.uint8)
    for b in range(B):
        for q in range(Q):
            if q % 17 == 3:
                continue                                              # empty masks: never emitted
            y0, x0 = rng.integers(0, H - 8), rng.integers(0, W - 8)
            hh, ww = rng.integers(4, 24), rng.integers(4, 32)
            masks[b, q, y0:y0 + hh, x0:x0 + ww] = 1
        masks[b, 10] = masks[b, 11]                                   # identical masks: IoU exactly 1
    scores = rng.random((B, Q)).astype(np.float32) * 0.9 + 0.05
    scores[:, 20] = 0.0009                                            # below the keep threshold once multiplied
    cats = rng.integers(0, 6, (B, Q)).astype(np.int64)                # class 0 = background: skipped
    cats[1] = 2                                                       # one image with a single crowded class
    # COCO-range ids that wrap in CPython's 32-slot set table (33 -> slot 1, 40 -> slot 8): the reference iterates
    # set(category_ids) (zutis.py:237-238), which is NOT ascending here — [40, 33, 2, 3] for {33, 2, 40, 3}
    cats[2] = np.array([33, 2, 40, 3, 80, 65], np.int64)[rng.integers(0, 6, Q)]
    assert [int(c) for c in set(cats[2])] != sorted(set(int(c) for c in cats[2]))
    eng_kept = ZutisEngine.instance_nms(None, torch.from_numpy(masks).to(dev), torch.from_numpy(scores).to(dev),
                                        torch.from_numpy(cats).to(dev), nms_type)
    ref = []
    for b in range(B):
        ref += [(b, c, q, s) for (c, q, s) in O.mask_nms(masks[b].astype(bool), scores[b], cats[b], nms_type)]
    assert [(b, c, q) for b, c, q, _ in eng_kept] == [(b, c, q) for b, c, q, _ in ref]
    got_s, ref_s = np.array([s for *_, s in eng_kept]), np.array([s for *_, s in ref])
    if nms_type == "hard":
        assert np.array_equal(got_s, ref_s)
    else:
        assert np.abs(got_s - ref_s).max() < 1e-12
    assert len(eng_kept) > 20
    # the chained form the drop-in's predict uses (NMS -> run extraction from the loop's device outputs, one host round trip): the
    # same kept list, and RLE / boxes / areas identical to the two-step path (instance_nms -> host -> encode_masks)
    md = torch.from_numpy(masks).to(dev)
    flag = torch.zeros((1,), dtype=torch.int32, device=dev)
    kept2, rles2, boxes2, areas2, bad = ZutisEngine.instance_nms_encode(None, md, torch.from_numpy(scores).to(dev), torch.from_numpy(cats).to(dev),
                                                                        nms_type, range_flag=flag)
    assert not bad and kept2 == eng_kept
    sel = np.array([b * Q + q for b, _, q, _ in eng_kept], dtype=np.int32)
    rles1, boxes1, areas1 = ZutisEngine.encode_masks(None, md.view(B * Q, H, W), sel)
    assert rles2 == rles1 and boxes2 == boxes1 and areas2 == areas1
    # the kept masks' run positions ride along with the small tables as ONE packed list: a head too short for them (second copy of the
    # whole list) and a run capacity below some masks' transitions (those are re-encoded from the mask) give the same strings
    for kw in (dict(pack_head=8), dict(max_runs=3), dict(max_runs=3, pack_head=8), dict(fused=False), dict(fused=False, pack_head=8),
               dict(fused=False, max_runs=3), dict(fused=True)):
        kept3, rles3, boxes3, areas3, _ = ZutisEngine.instance_nms_encode(None, md, torch.from_numpy(scores).to(dev), torch.from_numpy(cats).to(dev),
                                                                          nms_type, **kw)
        assert kept3 == eng_kept and rles3 == rles1 and boxes3 == boxes1 and areas3 == areas1, kw


def test_device_rle_strings_equal_the_host_encoder(dev):
    """zh_mask_rle_kept: the COCO RLE strings the device writes from the packed run list are byte for byte the host encoder's
    (= pycocotools.mask.encode, tests/test_rle.py pins that one): noisy masks (long strings, several 256-run chunks), blobs, an empty and
    a full mask, pixel 0 set, a single pixel at the very end; a mask over max_runs and lists past the capacity report -1."""
    from zutis_amd import ops, rle
    rng = np.random.default_rng(11)
    B, Q, H, W = 2, 9, 61, 83
    masks = np.zeros((B, Q, H, W), np.uint8)
    for b in range(B):
        masks[b, 0] = rng.random((H, W)) > 0.5                      # ~2500 transitions
        masks[b, 1, 10:40, 5:70] = 1
        masks[b, 3] = 1                                             # full (2: empty)
        masks[b, 4, 0, 0] = 1
        masks[b, 5, H - 1, W - 1] = 1
        masks[b, 6] = rng.random((H, W)) > 0.97
        masks[b, 7, :, ::2] = 1                                     # runs of exactly H pixels: every delta against run k - 2 is 0
        masks[b, 8, ::2, :] = 1                                     # runs of one pixel
    md = torch.from_numpy(masks).to(dev)
    order = [np.array([3, 0, 8, 2, 5, 1, 7], np.int32), np.array([4, 6, 0], np.int32)]
    idx = torch.zeros((B, Q), dtype=torch.int32)
    for b in range(B):
        idx[b, :len(order[b])] = torch.from_numpy(order[b])
    cnt = torch.tensor([len(o) for o in order], dtype=torch.int32)
    idx, cnt = idx.to(dev), cnt.to(dev)
    for max_runs, cap in ((8192, 1 << 16), (3000, 1 << 16), (8192, 6000)):
        pos = torch.empty((cap,), dtype=torch.int32, device=dev)
        nr = torch.empty((B * Q, 2), dtype=torch.int32, device=dev)
        ba = torch.empty((B * Q, 5), dtype=torch.int32, device=dev)
        ops.mask_runs_kept(md, idx, cnt, max_runs, pos, nr, ba, packed=True)
        out = torch.zeros((5 * cap + 16 * B * Q,), dtype=torch.uint8, device=dev)
        ln = torch.full((B * Q,), -7, dtype=torch.int32, device=dev)
        ops.mask_rle_kept(pos, nr, cnt, B, Q, max_runs, H * W, out, ln)
        out_h, ln_h, nr_h = out.cpu().numpy(), ln.cpu().numpy().reshape(B, Q), nr.cpu().numpy().reshape(B, Q, 2)
        off = rank = n_bad = 0
        for b in range(B):
            for j, q in enumerate(order[b]):
                want = rle.encode(masks[b, q])["counts"]
                nt = int(nr_h[b, j, 0])
                fits = nt <= max_runs and off + nt <= cap
                if fits:
                    c0 = 5 * off + 16 * rank
                    assert ln_h[b, j] == len(want) and out_h[c0:c0 + len(want)].tobytes() == want, (max_runs, cap, b, j)
                else:
                    assert ln_h[b, j] == -1, (max_runs, cap, b, j)
                    n_bad += 1
                off += min(nt, max_runs)
                rank += 1
            assert (ln_h[b, len(order[b]):] == -7).all()            # slots past the count are not touched
        assert (n_bad > 0) == ((max_runs, cap) != (8192, 1 << 16))


@pytest.mark.parametrize("H,W", [(61, 83), (48, 64), (37, 200), (120, 1024)])
def test_fused_run_box_string_kernel(dev, H, W):
    """zh_mask_rle_fused_kept (one workgroup per kept mask: bits in LDS, word-parallel transitions, string placed by an atomic cursor):
    strings byte for byte the host encoder's, boxes / areas = numpy's, for widths that are / are not multiples of 64 and pixel counts
    that are / are not multiples of 16; a mask over max_runs and strings past the capacity report length -1 with box and area intact."""
    from zutis_amd import ops, rle
    assert ops.mask_rle_fused_supported(H, W, 8192) and not ops.mask_rle_fused_supported(64, 1025, 8192)
    rng = np.random.default_rng(H * 1000 + W)
    B, Q = 2, 10
    masks = np.zeros((B, Q, H, W), np.uint8)
    for b in range(B):
        masks[b, 0] = rng.random((H, W)) > 0.5
        masks[b, 1, H // 5:H // 2, W // 7:W - 3] = 1
        masks[b, 3] = 1                                             # full (2: empty)
        masks[b, 4, 0, 0] = 1
        masks[b, 5, H - 1, W - 1] = 1
        masks[b, 6] = (rng.random((H, W)) > 0.97) * 255             # any non-zero byte counts as set
        masks[b, 7, :, ::2] = 1
        masks[b, 8, ::2, :] = 1
        masks[b, 9, :, W - 1] = 1                                   # last column only: the row-0 rule across the column boundary
        masks[b, 9, H - 1, W - 2] = 1
    md = torch.from_numpy(masks).to(dev)
    order = [np.array([3, 0, 8, 2, 5, 1, 7, 9], np.int32), np.array([4, 6, 0, 9], np.int32)]
    idx = torch.zeros((B, Q), dtype=torch.int32)
    for b in range(B):
        idx[b, :len(order[b])] = torch.from_numpy(order[b])
    cnt = torch.tensor([len(o) for o in order], dtype=torch.int32)
    idx, cnt = idx.to(dev), cnt.to(dev)
    want = {(b, j): rle.encode((masks[b, q] != 0).astype(np.uint8))["counts"] for b in range(B) for j, q in enumerate(order[b])}
    bits = torch.empty((B, Q, (H * W + 63) // 64), dtype=torch.int64, device=dev)       # the IoU step's bit-packed masks: the second source
    for b in range(B):
        ops.mask_iou_counts(md[b], Q, H * W, torch.empty((Q, Q), dtype=torch.int32, device=dev), torch.empty((Q, Q), dtype=torch.int32, device=dev),
                            workspace=bits[b])
    for max_runs, cap, src in ((8192, 1 << 20, None), (8192, 1 << 20, bits), (40, 1 << 20, bits), (8192, max(len(v) for v in want.values()) + 5, None)):
        out = torch.zeros((cap,), dtype=torch.uint8, device=dev)
        cursor = torch.zeros((1,), dtype=torch.int32, device=dev)
        info = torch.full((B * Q, 8), -7, dtype=torch.int32, device=dev)
        ops.mask_rle_fused_kept(md, idx, cnt, max_runs, out, cursor, info, bits=src)
        out_h, info_h = out.cpu().numpy(), info.cpu().numpy().reshape(B, Q, 8)
        n_bad, spans = 0, []
        for b in range(B):
            for j, q in enumerate(order[b]):
                mk = masks[b, q] != 0
                c0, ln, x0, y0, x1, y1, ar, nt = info_h[b, j].tolist()
                f = mk.reshape(-1, order="F")
                assert nt == int((f[1:] != f[:-1]).sum()) and ar == int(mk.sum()), (H, W, max_runs, cap, b, j)
                if ar:
                    ys, xs = np.nonzero(mk)
                    assert (x0, y0, x1, y1) == (xs.min(), ys.min(), xs.max(), ys.max())
                else:
                    assert (x0, y0, x1, y1) == (W, H, -1, -1)
                if ln >= 0:
                    assert nt <= max_runs and out_h[c0:c0 + ln].tobytes() == want[b, j], (H, W, max_runs, cap, b, j)
                    spans.append((c0, c0 + ln))
                else:
                    assert nt > max_runs or cap < (1 << 20), (H, W, max_runs, cap, b, j)     # with room for every string only the run limit refuses
                    n_bad += 1
            assert (info_h[b, len(order[b]):] == -7).all()          # slots past the count are not touched
        spans.sort()
        assert all(a1 <= b0 for (_, a1), (b0, _) in zip(spans, spans[1:]))      # the cursor hands out disjoint places
        assert len(spans) > 0 and (n_bad > 0 or (max_runs, cap) == (8192, 1 << 20))


def test_range_flag_of_the_instance_statistics(dev):
    """The reference asserts 0 <= mask_proposals <= 1 (zutis.py:385-386): zh_instance_mask_stats raises a device flag for a value
    outside the range or a NaN, and leaves it alone otherwise."""
    from zutis_amd import ops
    B, Q, M = 2, 7, 300
    mp = torch.rand((B, Q, M), generator=torch.Generator().manual_seed(3))
    for bad, val in ((False, None), (True, 1.0001), (True, -1e-6), (True, float("nan"))):
        x = mp.clone()
        if bad:
            x[1, 5, 299] = val
        flag = torch.zeros((1,), dtype=torch.int32, device=dev)
        sizes, conf = torch.empty((B * Q,), device=dev), torch.empty((B * Q,), device=dev)
        binary = torch.empty((B, Q, M), dtype=torch.uint8, device=dev)
        ops.instance_mask_stats(x.to(dev), Q * M, 0.5, B, Q, M, sizes, conf, binary, flag)
        assert bool(flag.item()) == bad
        assert torch.equal(binary.cpu(), (x > 0.5).to(torch.uint8))


@pytest.mark.parametrize("h,w,H,W", [(120, 160, 480, 640), (107, 160, 427, 640), (30, 40, 123, 164), (21, 21, 336, 336), (9, 13, 50, 1030)])
def test_thresholded_mask_upsample_rows_kernel_is_bitwise_the_flat_kernel(dev, h, w, H, W):
    """Instance masks (zutis.py:422-423: F.interpolate(mask_proposals, size, 'bilinear') > threshold): the mask-only call takes the
    row-wise kernel (one block row per output row, 4 pixels per store, W % 4 == 0); it must equal the flat kernel's float output
    compared with the threshold, pixel for pixel, and ATen's interpolate (values within an ulp of the threshold may differ there)."""
    import torch.nn.functional as F
    from zutis_amd import ops
    planes, thr = 7, 0.5
    x = torch.rand((planes, h, w), generator=torch.Generator().manual_seed(h * W)).to(dev)
    out = torch.empty((planes, H, W), dtype=torch.float32, device=dev)
    m_flat = torch.empty((planes, H, W), dtype=torch.uint8, device=dev)
    ops.upsample_bilinear_nchw(x, planes, h, w, H, W, out=out, mask_u8=m_flat, threshold=thr)       # float output requested: flat kernel
    m_rows = torch.full((planes, H, W), 7, dtype=torch.uint8, device=dev)
    ops.upsample_bilinear_nchw(x, planes, h, w, H, W, mask_u8=m_rows, threshold=thr)                # mask only: row-wise kernel
    assert torch.equal(m_rows, m_flat)
    assert torch.equal(m_rows.bool(), out > thr)
    ref = F.interpolate(x[None].cpu(), size=(H, W), mode="bilinear")[0]
    assert torch.equal(out.cpu(), ref)                                                               # ATen-exact arithmetic


def test_attention_x3_two_query_tiles_per_wave_is_bitwise_the_product_kernel(dev, tmp_path):
    """Developer variant ZH_ATTN_QT=2 (round 5: a wave owns two 32-query tiles and every K / V fragment read feeds both; measured a tie
    with the product kernel at the headline's shape, profiles/NOTES.md): the same arithmetic per query, so the outputs agree bit for bit —
    ragged T (tile B of the last wave beyond Tq), several heads and images.  The switch is read once per process: the variant runs in a
    child process on the tensors this one wrote."""
    import os, subprocess, sys
    from zutis_amd import ops
    from zutis_amd.ops import Act
    B, H, dh, T = 3, 4, 64, 333
    D = H * dh
    x = [_randn((B * T, D), 900 + i, 1.0).to(dev) for i in range(3)]
    torch.save([t.cpu() for t in x], tmp_path / "qkv.pt")
    script = tmp_path / "child.py"
    script.write_text(
        "import sys, torch\n"
        f"sys.path.insert(0, {repr(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))})\n"
        "from zutis_amd import ops\nfrom zutis_amd.ops import Act\n"
        f"B, H, dh, T, D = {B}, {H}, {dh}, {T}, {D}\n"
        "dev = torch.device('cuda:0')\n"
        f"x = [t.to(dev) for t in torch.load({repr(str(tmp_path / 'qkv.pt'))})]\n"
        "def pair(t):\n    a = Act.empty(tuple(t.shape), True, dev); ops.cast_f16(t, a, t.shape[0], t.shape[1]); return a\n"
        "q, k, v = (pair(t) for t in x)\no = Act.empty((B * T, D), True, dev)\n"
        "ops.attention(q, k, v, o, batch=B, heads=H, Tq=T, Tk=T, head_dim=dh, ldq=D, ldk=D, ldv=D, ldo=D, strideQ=T * D, strideK=T * D, strideV=T * D, strideO=T * D, x3=True)\n"
        f"torch.save(o.t.cpu(), {repr(str(tmp_path / 'o.pt'))})\n")

    def pair(t):
        a = Act.empty(tuple(t.shape), True, dev)
        ops.cast_f16(t, a, t.shape[0], t.shape[1])
        return a
    q, k, v = (pair(t) for t in x)
    o = Act.empty((B * T, D), True, dev)
    ops.attention(q, k, v, o, batch=B, heads=H, Tq=T, Tk=T, head_dim=dh, ldq=D, ldk=D, ldv=D, ldo=D, strideQ=T * D, strideK=T * D, strideV=T * D,
                  strideO=T * D, x3=True)
    r = subprocess.run([sys.executable, str(script)], env={**os.environ, "ZH_ATTN_QT": "2"}, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert torch.equal(torch.load(tmp_path / "o.pt"), o.t.cpu())
