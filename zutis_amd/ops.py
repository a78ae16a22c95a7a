"""Thin torch-tensor -> raw-pointer wrappers over the C ABI (include/zutis_hip.h).

PyTorch is plumbing here: device memory, the current HIP stream.  Every op launches a hand-written HIP
kernel from libzutis_hip.so; a missing library or a non-GPU tensor is an error, never a fallback.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch

from . import _lib

ACT_NONE, ACT_QUICKGELU, ACT_RELU, ACT_SIGMOID, ACT_GELU_ERF = 0, 1, 2, 3, 4
f16, f32 = torch.float16, torch.float32


# Optional per-launch profiler (bench.py): PROFILER(name, work, launch) must call launch() and may bracket it with
# HIP events on the current stream.  None in production: zero overhead.
PROFILER = None


def _launch(name, work, fn):
    if PROFILER is None:
        return fn()
    return PROFILER(name, work, fn)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    """Raw handle of torch's current HIP stream.  torch.cuda.current_stream() costs ~8 us of Python per call (half of the
    per-launch host overhead, tools/py_overhead.py); the private raw getter is ~0.3 us and returns the same handle."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.ZutisHipError("zutis_amd ops need GPU tensors (no CPU fallback)")
    if _raw_device is not None and t.device.index != _raw_device():
        raise _lib.ZutisHipError(f"tensor on cuda:{t.device.index} but the current device (whose stream is used) is "
                                 f"cuda:{_raw_device()}: wrap the call in torch.cuda.device(...)")
    if _lib.RECORDER is not None:
        _lib.RECORDER.keepalive.append(t)        # a launch plan owns every tensor whose address it recorded
    return t.data_ptr()


def _chk(t: torch.Tensor, dtype, name: str):
    if t.dtype != dtype or not t.is_contiguous():
        raise _lib.ZutisHipError(f"{name}: expected contiguous {dtype}, got {t.dtype} contiguous={t.is_contiguous()}")


class Act:
    """An fp16 activation (or weight) tensor, optionally stored as a split pair for the f16x3 precision (zutis_hip.h):
    `t` has shape [P, ...] with P = 1 (plain fp16) or 2 (hi, lo); `hi` = t[0] is the plain fp16 tensor every fp16 consumer
    reads; `plane` = element offset hi -> lo (0 when not split); `out_scale` = 2^-s for weights packed as W * 2^s; `x2`: a
    split_weight() pack whose lo plane would be all zeros, stored as its hi plane alone."""
    __slots__ = ("t", "hi", "plane", "out_scale", "x2")

    def __init__(self, t: torch.Tensor, out_scale: float = 1.0):
        assert t.dtype == f16 and t.shape[0] in (1, 2)
        self.t, self.hi = t, t[0]
        self.plane = t.stride(0) if t.shape[0] == 2 else 0
        self.out_scale = float(out_scale)
        self.x2 = False      # set by split_weight(): one plane that IS the fp32 weight (times 2^s) exactly — the f16x2 operand form

    @staticmethod
    def empty(shape, split: bool, device) -> "Act":
        return Act(torch.empty((2 if split else 1,) + tuple(shape), dtype=f16, device=device))

    def view(self, t_hi: torch.Tensor) -> "Act":
        """The same pair seen through a view of the hi plane (column / row slices keep the plane offset)."""
        a = Act.__new__(Act)
        a.t, a.hi, a.plane, a.out_scale, a.x2 = self.t, t_hi, self.plane, self.out_scale, self.x2
        return a


ALLOW_X2 = True     # developer switch (tests / A-B): False packs every weight as two planes, zero lo plane or not


def split_weight(w32: torch.Tensor, allow_x2: Optional[bool] = None) -> Act:
    """Pack-time split of an fp32 weight for zh_gemm_f16x3: W * 2^s with s chosen so that max|W| lands in [2^13, 2^14)
    (hi far from overflow, lo = f16(W*2^s - hi) a normal fp16 number); out_scale = 2^-s is exact.  A weight whose lo plane
    would be all zeros is packed as ONE plane (the "f16x2" form of zh_gemm_f16x3) unless allow_x2 (default: ALLOW_X2) is False."""
    w = w32.detach().to(f32)
    m = float(w.abs().max()) if w.numel() else 0.0
    s = 0 if m == 0.0 or not math.isfinite(m) else 13 - math.floor(math.log2(m))
    s = max(-14, min(60, s))                        # 2^-60 is still a normal fp32 out_scale
    ws = w * (2.0 ** s)
    hi = ws.to(f16)
    lo = (ws - hi.to(f32)).to(f16)
    if (ALLOW_X2 if allow_x2 is None else allow_x2) and not bool(lo.any()):
        # every value of W * 2^s is an fp16 number: the released CLIP towers, whose Linear / conv / attention weights the
        # reference's constructor rounds to fp16 (convert_weights, clip_arch.py:566-587,625) before they are cast back to fp32
        # (zutis.py:55).  One plane, plane = 0: zh_gemm_f16x3 then skips the product with the zero lo plane (bit-identical).
        a = Act(hi.unsqueeze(0).contiguous(), out_scale=2.0 ** -s)
        a.x2 = True
        return a
    return Act(torch.stack([hi, lo]).contiguous(), out_scale=2.0 ** -s)


def _hp(a):
    """(hi tensor, lo-plane offset) of an Act or a plain fp16 tensor."""
    return (a.hi, a.plane) if isinstance(a, Act) else (a, 0)


def _gemm_bytes(M, N, K, batch, strideA, strideW, op_bytes, out_bytes, has_residual) -> float:
    """Compulsory (algorithmic) bytes of one GEMM launch: each operand once (a batch-shared operand once), the output once,
    the fp32 residual once.  op_bytes = 2 for fp16 operands, 4 for split pairs."""
    a = M * K * op_bytes * (batch if strideA else 1)
    w = N * K * op_bytes * (batch if (strideW or batch == 1) else 1)
    return float(a + w + M * N * batch * (out_bytes + (4 if has_residual else 0)))


def _pos(pos, N):
    """Argument words of the optional separable row bias: pos = (pos_y [h, >=N], pos_x [w, >=N]), both fp32 or both fp16 ->
    rows are pixels m = img * h*w + y * w + x and pos_y[y] + pos_x[x] is added before the activation."""
    if pos is None:
        return (None, None, 0, 0, 0, 0)
    ty, tx = pos
    assert ty.dtype == tx.dtype and ty.dtype in (f32, f16) and ty.dim() == 2 and tx.dim() == 2 and ty.stride(1) == 1 and tx.stride(1) == 1
    assert ty.stride(0) == tx.stride(0) and ty.shape[1] >= N and tx.shape[1] >= N
    return (_p(ty), _p(tx), ty.stride(0), ty.shape[0], tx.shape[0], int(ty.dtype == f16))


def gemm_x3(A: Act, W: Act, out, bias=None, residual=None, res_rows: int = 0, act: int = ACT_NONE, *, M=None, N=None, K=None,
            lda=None, ldw=None, ldc=None, batch: int = 1, strideA: int = 0, strideW: int = 0, strideC: int = 0, ldr=None,
            strideR: int = 0, pos=None, fixed_k_order: bool = False):
    """fixed_k_order: results of calls with different M / N are compared bit for bit (sharded retrieval): ring kernels only.
    out = act((A @ W^T) * W.out_scale + bias + pos) + residual[m % res_rows] at the reference's fp32-class precision: A and W
    are split pairs (Act with plane != 0); out is an f32 tensor, an fp16 tensor / plain Act, or a split Act (then the residual is
    added in fp32 before the one rounding, act must be none).  pos: see _pos()."""
    L = _lib.load()
    if not (isinstance(A, Act) and isinstance(W, Act) and A.plane and (W.plane or W.x2)):
        raise _lib.ZutisHipError("gemm_x3: A must be a split pair, W a split pair or a one-plane split_weight() pack")
    a, w = A.hi, W.hi
    M = a.shape[-2] if M is None else M
    K = a.shape[-1] if K is None else K
    N = w.shape[-2] if N is None else N
    lda = a.stride(-2) if lda is None else lda
    ldw = w.stride(-2) if ldw is None else ldw
    o, planeC = _hp(out)
    kind = 0 if o.dtype == f32 else (2 if planeC else 1)
    ldc = o.stride(-2) if ldc is None else ldc
    if residual is not None:
        assert residual.dtype == f32 and (kind == 0 or act == ACT_NONE)
        ldr = residual.stride(-2) if ldr is None else ldr
        res_rows = res_rows or M
    if bias is not None:
        assert bias.dtype == f32 and bias.numel() >= N
    args = (_p(a), lda, strideA, A.plane, _p(w), ldw, strideW, W.plane, _p(o), ldc, strideC, planeC, kind,
            float(A.out_scale * W.out_scale), _p(bias), _p(residual), ldr or 0, strideR, res_rows, *_pos(pos, N),
            act, M, N, K, batch, int(bool(fixed_k_order)), _stream())
    nbytes = _gemm_bytes(M, N, K, batch, strideA, strideW, 4, 4 if kind == 0 else (4 if kind == 2 else 2), residual is not None)
    if not W.plane:                                  # one-plane weight: 2 bytes per element instead of 4
        nbytes -= N * K * 2.0 * (batch if (strideW or batch == 1) else 1)
    name = "gemm_f16x3" if W.plane else "gemm_f16x2"
    _lib.check(_launch(name, (2.0 * M * N * K * batch, nbytes, (M, N, K, batch)), lambda: L.zh_gemm_f16x3(*args)), "zh_gemm_f16x3")
    return out


def gemm(A: torch.Tensor, W: torch.Tensor, out: torch.Tensor, bias=None, residual=None, res_rows: int = 0,
         act: int = ACT_NONE, *, M=None, N=None, K=None, lda=None, ldw=None, ldc=None, batch: int = 1,
         strideA: int = 0, strideW: int = 0, strideC: int = 0, ldr=None, strideR: int = 0, pos=None):
    """out = act(A @ W^T + bias + pos) + residual[m % res_rows].  A [M,K] f16, W [N,K] f16, out f32|f16 [M,N]; pos: see _pos().
    Act operands / outputs are read / written through their hi plane (plain fp16)."""
    L = _lib.load()
    for t in (A, W):
        if isinstance(t, Act) and t.out_scale != 1.0:   # a weight packed for the x3 mode is W * 2^s: its hi plane alone is not W
            raise _lib.ZutisHipError("gemm: operand packed with out_scale != 1 (split_weight) reached the fp16-operand GEMM")
    A, W, out_ret = _hp(A)[0], _hp(W)[0], out
    if isinstance(out, Act):
        if out.plane:
            raise _lib.ZutisHipError("gemm: an fp16-operand GEMM cannot fill a split-pair output (its lo plane would be stale)")
        out = out.hi
    M = A.shape[-2] if M is None else M
    K = A.shape[-1] if K is None else K
    N = W.shape[-2] if N is None else N
    lda = A.stride(-2) if lda is None else lda
    ldw = W.stride(-2) if ldw is None else ldw
    ldc = out.stride(-2) if ldc is None else ldc
    assert A.dtype == f16 and W.dtype == f16 and out.dtype in (f16, f32)
    if residual is not None:
        assert residual.dtype == f32
        ldr = residual.stride(-2) if ldr is None else ldr
        res_rows = res_rows or M
    if bias is not None:
        assert bias.dtype == f32 and bias.numel() >= N
    args = (_p(A), lda, strideA, _p(W), ldw, strideW, _p(out), ldc, strideC, int(out.dtype == f16),
            _p(bias), _p(residual), ldr or 0, strideR, res_rows, *_pos(pos, N), act, M, N, K, batch, _stream())
    nbytes = _gemm_bytes(M, N, K, batch, strideA, strideW, 2, 2 if out.dtype == f16 else 4, residual is not None)
    _lib.check(_launch("gemm_f16", (2.0 * M * N * K * batch, nbytes, (M, N, K, batch)), lambda: L.zh_gemm_f16(*args)), "zh_gemm_f16")
    return out_ret


def attention_splitk_workspace_size(batch, heads, Tq, head_dim, ksplit) -> int:
    return int(_lib.load(raw=True).zh_attention_splitk_workspace_size(batch, heads, Tq, head_dim, ksplit))


def attention(Q, K, V, O, *, batch, heads, Tq, Tk, head_dim, ldq, ldk, ldv, ldo, strideQ, strideK, strideV, strideO,
              scale=None, causal=False, x3=False, ksplit: int = 1, workspace=None):
    """x3: Q, K and V are split-pair Acts; scores and P.V get the three-product fp32-class form; a split O is filled as a pair.
    ksplit > 1: the keys are split over ksplit workgroups per (image, head, query block) and merged by a second launch; workspace =
    a uint8 CUDA tensor of attention_splitk_workspace_size() bytes."""
    L = _lib.load()
    scale = 1.0 / math.sqrt(head_dim) if scale is None else scale
    (Q, pq), (K, pk), (V, pv), (O, po) = _hp(Q), _hp(K), _hp(V), _hp(O)
    if x3 and not (pq and pk and pv):
        raise _lib.ZutisHipError("attention(x3): Q, K and V must be split pairs")
    if not x3:
        pq = pk = pv = 0
    name = "attention_f16x3" if x3 else "attention_f16"
    if causal:
        if Tq != Tk:
            raise _lib.ZutisHipError("causal attention needs Tq == Tk")
        args = (_p(Q), ldq, strideQ, _p(K), ldk, strideK, _p(V), ldv, strideV, _p(O), ldo, strideO,
                batch, heads, Tq, head_dim, float(scale), pq, pk, pv, po, _stream())
        _lib.check(_launch(name, 2.0 * batch * heads * Tq * Tk * head_dim, lambda: L.zh_attention_causal_f16(*args)),
                   "zh_attention_causal_f16")
        return O
    if ksplit > 1:
        need = attention_splitk_workspace_size(batch, heads, Tq, head_dim, ksplit)
        if workspace is None or workspace.numel() * workspace.element_size() < need:
            raise _lib.ZutisHipError(f"attention(ksplit={ksplit}): workspace of {need} bytes required")
        args = (_p(Q), ldq, strideQ, _p(K), ldk, strideK, _p(V), ldv, strideV, _p(O), ldo, strideO,
                batch, heads, Tq, Tk, head_dim, float(scale), pq, pk, pv, po, ksplit, _p(workspace),
                workspace.numel() * workspace.element_size(), _stream())
        _lib.check(_launch(name, 4.0 * batch * heads * Tq * Tk * head_dim, lambda: L.zh_attention_f16_splitk(*args)),
                   "zh_attention_f16_splitk")
        return O
    args = (_p(Q), ldq, strideQ, _p(K), ldk, strideK, _p(V), ldv, strideV, _p(O), ldo, strideO,
            batch, heads, Tq, Tk, head_dim, float(scale), pq, pk, pv, po, _stream())
    _lib.check(_launch(name, 4.0 * batch * heads * Tq * Tk * head_dim, lambda: L.zh_attention_f16(*args)),
               "zh_attention_f16")
    return O


def embed_tokens(tokens, table, pos, out):
    """tokens int64 [n,ctx] -> out f32 [n*ctx, D] = table[tokens] + pos (clip_arch.py:535-537)."""
    L = _lib.load()
    n, ctx = tokens.shape
    _chk(table, f32, "token table"); _chk(pos, f32, "positional embedding"); _chk(out, f32, "embed out")
    if tokens.dtype != torch.int64 or not tokens.is_contiguous():
        raise _lib.ZutisHipError("embed_tokens: tokens must be contiguous int64")
    _lib.check(L.zh_embed_tokens_f32(_p(tokens), _p(table), _p(pos), _p(out), n, ctx, table.shape[1], table.shape[0], _stream()),
               "zh_embed_tokens_f32")


def eot_rows(tokens, x, out):
    L = _lib.load()
    n, ctx = tokens.shape
    _chk(x, f32, "eot x"); _chk(out, f32, "eot out")
    _lib.check(L.zh_eot_rows_f32(_p(tokens), _p(x), _p(out), n, ctx, out.shape[1], _stream()), "zh_eot_rows_f32")


def group_mean_l2norm(x, out, groups, T, E):
    L = _lib.load()
    _chk(x, f32, "group mean x"); _chk(out, f32, "group mean out")
    _lib.check(L.zh_group_mean_l2norm(_p(x), _p(out), groups, T, E, _stream()), "zh_group_mean_l2norm")


STATUS_RANGE, STATUS_NONFINITE = 1, 2        # bits of the status word (zutis_hip.h ZH_STATUS_*)
UNIT_NORM_SCALE = 1024.0                     # f16_scale of the unit-norm producers (2^10): see zutis_hip.h


def layernorm(x, gamma, beta, eps, rows, D, *, out_f32=None, out_f16=None, out_f16_plus=None, out_f32_plus=None,
              add=None, add_rows=0, in_group_rows=None, in_group_stride=None, in_offset=0,
              out_group_rows=None, out_group_stride=None, out_offset=0, status=None):
    L = _lib.load()
    in_group_rows = rows if in_group_rows is None else in_group_rows
    in_group_stride = in_group_rows if in_group_stride is None else in_group_stride
    out_group_rows = rows if out_group_rows is None else out_group_rows
    out_group_stride = out_group_rows if out_group_stride is None else out_group_stride
    (out_f16, p1), (out_f16_plus, p2) = _hp(out_f16), _hp(out_f16_plus)
    if out_f16 is not None and out_f16_plus is not None and p1 != p2:
        raise _lib.ZutisHipError("layernorm: both fp16 outputs must be split pairs of the same shape, or both plain")
    lo_plane = p1 or p2
    _lib.check(L.zh_layernorm_f32(_p(x), in_group_rows, in_group_stride, in_offset,
                                  out_group_rows, out_group_stride, out_offset, _p(gamma), _p(beta), float(eps),
                                  _p(out_f32), _p(out_f16), _p(out_f16_plus), _p(out_f32_plus), _p(add), add_rows,
                                  rows, D, lo_plane, _p(status), _stream()), "zh_layernorm_f32")


def sum_layernorm(parts, n_parts, rows, D, *, part_stride=None, bias=None, residual=None, out_sum=None, gamma=None, beta=None, eps=1e-5,
                  out_f32=None, out_f16=None, out_group_rows=None, out_group_stride=None, out_offset=0, skip_first_in_group=False,
                  gamma2=None, beta2=None, eps2=1e-5, out2_f32=None, out2_f16=None, out2_group_rows=None, out2_group_stride=None, out2_offset=0,
                  status=None):
    """x = sum of the n_parts fp32 planes of `parts` + bias + residual -> out_sum; LN(x) -> out_f32 / out_f16 (row-mapped); LN(LN(x)) with
    gamma2 / beta2 -> out2_* (zh_sum_layernorm_f32).  n_parts = 1 with no bias / residual is a plain (or chained) LayerNorm."""
    L = _lib.load()
    part_stride = rows * D if part_stride is None else part_stride
    ogr = rows if out_group_rows is None else out_group_rows
    ogs = ogr if out_group_stride is None else out_group_stride
    ogr2 = rows if out2_group_rows is None else out2_group_rows
    ogs2 = ogr2 if out2_group_stride is None else out2_group_stride
    (out_f16, lo1), (out2_f16, lo2) = _hp(out_f16), _hp(out2_f16)
    _lib.check(L.zh_sum_layernorm_f32(_p(parts), n_parts, part_stride, _p(bias), _p(residual), _p(out_sum), _p(gamma), _p(beta), float(eps),
                                      _p(out_f32), _p(out_f16), lo1, ogr, ogs, out_offset, int(skip_first_in_group),
                                      _p(gamma2), _p(beta2), float(eps2), _p(out2_f32), _p(out2_f16), lo2, ogr2, ogs2, out2_offset,
                                      rows, D, _p(status), _stream()), "zh_sum_layernorm_f32")


def assemble_tokens_ln(patch_emb, cls, pos, gamma, beta, eps, out, B, T, D):
    L = _lib.load()
    _lib.check(L.zh_assemble_tokens_ln(_p(patch_emb), _p(cls), _p(pos), _p(gamma), _p(beta), float(eps), _p(out),
                                       B, T, D, _stream()), "zh_assemble_tokens_ln")


def _f16_scale(a) -> float:
    """The factor a producer stores into Act `a` with: 1 / a.out_scale (a plain tensor or an unscaled Act: 1)."""
    return 1.0 / a.out_scale if isinstance(a, Act) else 1.0


def l2norm_rows(x, rows, D, out_f32=None, out_f16=None, eps=0.0):
    """out_f16 may be an Act with out_scale = 2^-s: the fp16 / split-pair copy is then stored times 2^s (unit-norm rows: UNIT_NORM_SCALE)."""
    L = _lib.load()
    sc = _f16_scale(out_f16)
    out_f16, lo = _hp(out_f16)
    _lib.check(L.zh_l2norm_rows(_p(x), _p(out_f32), _p(out_f16), float(eps), rows, D, lo, sc, _stream()), "zh_l2norm_rows")


def global_ln_l2_workspace_size(B, M, Cc) -> int:
    return int(_lib.load(raw=True).zh_global_ln_l2_workspace_size(B, M, Cc))


def global_ln_l2(x, B, M, Cc, out_f32=None, out_f16=None, eps=1e-5, l2_eps=1e-7, workspace=None, status=None):
    L = _lib.load()
    need = global_ln_l2_workspace_size(B, M, Cc)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=x.device)
    sc = _f16_scale(out_f16)
    out_f16, lo = _hp(out_f16)
    _lib.check(L.zh_global_ln_l2(_p(x), _p(out_f32), _p(out_f16), float(eps), float(l2_eps), B, M, Cc, _p(workspace),
                                 workspace.numel() * workspace.element_size(), lo, sc, _p(status), _stream()), "zh_global_ln_l2")


def im2col(x, out, patch, Kpad, pad_to_patch=False):
    L = _lib.load()
    B, Cin, H, W = x.shape
    _chk(x, f32, "im2col x")
    out, lo = _hp(out)
    _lib.check(L.zh_im2col_f16(_p(x), _p(out), B, Cin, H, W, patch, Kpad, int(pad_to_patch), lo, _stream()), "zh_im2col_f16")


def posembed_bicubic(pos, out, grid, h, w, D, scale_h, scale_w, has_cls=True):
    L = _lib.load()
    _lib.check(L.zh_posembed_bicubic(_p(pos), _p(out), grid, h, w, D, float(np.float32(scale_h)), float(np.float32(scale_w)),
                                     int(has_cls), _stream()), "zh_posembed_bicubic")


def upsample2x_cl(x, B, h, w, D, out_f32=None, out_f16=None, relu=False):
    L = _lib.load()
    out_f16, lo = _hp(out_f16)
    _lib.check(L.zh_upsample2x_bilinear_cl(_p(x), _p(out_f32), _p(out_f16), B, h, w, D, lo, int(relu), _stream()),
               "zh_upsample2x_bilinear_cl")


def sine_pe(out, h, w, D, temperature=10000.0):
    L = _lib.load()
    _lib.check(L.zh_sine_pe(_p(out), h, w, D, float(temperature), _stream()), "zh_sine_pe")


def add_rowperiodic_f16(a, add, out, rows, D, add_rows):
    L = _lib.load()
    (a, la), (out, lo) = _hp(a), _hp(out)
    _lib.check(L.zh_add_rowperiodic_f16(_p(a), _p(add), _p(out), rows, D, add_rows, la, lo, _stream()), "zh_add_rowperiodic_f16")


def fill_f32(x, value=0.0):
    L = _lib.load()
    _chk(x, f32, "fill x")
    _lib.check(L.zh_fill_f32(_p(x), float(value), x.numel(), _stream()), "zh_fill_f32")


def cast_f16(x, out, rows, D, add=None, add_rows=0):
    L = _lib.load()
    sc = _f16_scale(out)
    out, lo = _hp(out)
    _lib.check(L.zh_cast_f32_f16(_p(x), _p(add), add_rows, _p(out), rows, D, lo, sc, _stream()), "zh_cast_f32_f16")


def lin_scale(in_size: int, out_size: int) -> float:
    """ATen area_pixel_compute_scale for size= calls: float32(in)/float32(out)."""
    return float(np.float32(in_size) / np.float32(out_size))


def upsample_argmax(logits_lo, labels, B, n, h, w, H, W):
    L = _lib.load()
    _lib.check(L.zh_upsample_argmax(_p(logits_lo), _p(labels), B, n, h, w, H, W, lin_scale(h, H), lin_scale(w, W), _stream()),
               "zh_upsample_argmax")


def upsample_bilinear_nchw(x, planes, h, w, H, W, out=None, mask_u8=None, threshold=0.5, scale_h=None, scale_w=None):
    """scale_* default to in/out (size= form); pass 1/scale_factor for the scale_factor form with a cropped output."""
    L = _lib.load()
    sh = lin_scale(h, H) if scale_h is None else float(np.float32(scale_h))
    sw = lin_scale(w, W) if scale_w is None else float(np.float32(scale_w))
    _lib.check(L.zh_upsample_bilinear_nchw(_p(x), _p(out), _p(mask_u8), float(threshold), planes, h, w, H, W,
                                           sh, sw, _stream()), "zh_upsample_bilinear_nchw")


def confusion_hist(label_true, label_pred, hist, n_class):
    L = _lib.load()
    _chk(label_true, torch.int64, "label_true")
    _chk(label_pred, torch.int64, "label_pred")
    _chk(hist, torch.int64, "hist")
    _lib.check(L.zh_confusion_hist(_p(label_true), _p(label_pred), _p(hist), label_true.numel(), n_class, _stream()),
               "zh_confusion_hist")


def instance_mask_stats(mask_proposals_last, stride_image, threshold, B, Q, M, sizes, conf, binary, range_flag=None):
    """range_flag: int32 [1] (zeroed by the caller): bit 0 set when a proposal lies outside [0, 1] (zutis.py:385-386)."""
    L = _lib.load()
    _lib.check(L.zh_instance_mask_stats(_p(mask_proposals_last), stride_image, float(threshold), B, Q, M, _p(sizes), _p(conf),
                                        _p(binary), _p(range_flag), _stream()), "zh_instance_mask_stats")


def masked_mean_tokens(tokens, binary, sizes, avg, B, Q, M, E):
    L = _lib.load()
    need = int(_lib.load(raw=True).zh_masked_mean_workspace_size(B, Q, M, E))
    ws = torch.empty(need, dtype=torch.uint8, device=tokens.device)
    _lib.check(L.zh_masked_mean_tokens(_p(tokens), _p(binary), _p(sizes), _p(avg), B, Q, M, E, _p(ws), need, _stream()), "zh_masked_mean_tokens")


def instance_classify(avg, text, conf, temperature, rows, n, E, category, score):
    L = _lib.load()
    _lib.check(L.zh_instance_classify(_p(avg), _p(text), _p(conf), float(temperature), rows, n, E, _p(category), _p(score),
                                      _stream()), "zh_instance_classify")


def mask_iou_counts(masks_u8, n, pixels, inter, uni, workspace=None):
    """workspace (optional, >= zh_mask_iou_workspace_size bytes, any dtype): kept by the caller, it holds the masks bit-packed afterwards
    (u64 [n][(pixels + 63) // 64]: the `bits` of mask_rle_fused_kept)."""
    L = _lib.load()
    need = L.zh_mask_iou_workspace_size(n, pixels)
    ws = workspace if workspace is not None else torch.empty(need, dtype=torch.uint8, device=masks_u8.device)
    if ws.numel() * ws.element_size() < need:
        raise ZutisHipError(f"mask_iou_counts: workspace holds {ws.numel() * ws.element_size()} bytes, {need} needed")
    _lib.check(L.zh_mask_iou_counts(_p(masks_u8), n, pixels, _p(inter), _p(uni), _p(ws), need, _stream()), "zh_mask_iou_counts")


NMS_TYPES = {"hard": 0, "linear": 1, "gaussian": 2}


def mask_nms(inter, uni, scores, category_ids, nms_type="hard", nms_threshold=0.3, sigma=0.5, score_threshold=0.001, packed=None,
             range_flag=None, zero_word=None):
    """Greedy per-category mask NMS on the device (zutis.py:211-299).  inter / uni int32 [B,Q,Q], scores f32 [B,Q], category_ids
    int64 [B,Q] -> (index int32 [B,Q], score f64 [B,Q], category int64 [B,Q], count int32 [B]); the first count[b] entries of
    row b are the kept queries in the reference's emission order."""
    L = _lib.load()
    B, Q = scores.shape
    _chk(inter, torch.int32, "nms inter"); _chk(uni, torch.int32, "nms union"); _chk(scores, f32, "nms scores")
    _chk(category_ids, torch.int64, "nms categories")
    if nms_type not in NMS_TYPES:
        raise AssertionError(nms_type)                     # reference: assert nms_type in ["hard", "linear", "gaussian"]
    dev = scores.device
    idx = torch.empty((B, Q), dtype=torch.int32, device=dev)
    sc = torch.empty((B, Q), dtype=torch.float64, device=dev)
    cat = torch.empty((B, Q), dtype=torch.int64, device=dev)
    cnt = torch.empty((B,), dtype=torch.int32, device=dev)
    _lib.check(L.zh_mask_nms(_p(inter), _p(uni), _p(scores), _p(category_ids), B, Q, NMS_TYPES[nms_type], float(nms_threshold),
                             float(sigma), float(score_threshold), _p(idx), _p(sc), _p(cat), _p(cnt), _p(packed), _p(range_flag),
                             None if zero_word is None else _p(zero_word), _stream()), "zh_mask_nms")
    return idx, sc, cat, cnt


def mask_runs_kept(masks_u8, kept_index, kept_count, max_runs, pos, nr, ba, packed=False):
    """masks u8 [B,Q,H,W]; kept_index int32 [B,Q] / kept_count int32 [B] = zh_mask_nms' device outputs -> pos int32 [B*Q,max_runs], nr
    int32 [B*Q,2], ba int32 [B*Q,5] (row b*Q + j = image b's j-th kept mask; rows past the count are not written).  packed: pos is ONE
    int32 list (any length) that takes the kept masks' transitions back to back, min(#transitions, max_runs) each, as far as it reaches."""
    L = _lib.load()
    _chk(masks_u8, torch.uint8, "mask_runs_kept masks"); _chk(kept_index, torch.int32, "kept_index"); _chk(kept_count, torch.int32, "kept_count")
    _chk(pos, torch.int32, "mask_runs_kept pos")
    B, Q, H, W = masks_u8.shape
    if not packed and pos.numel() < B * Q * max_runs:
        raise ZutisHipError(f"mask_runs_kept: pos holds {pos.numel()} ints, the row form needs {B * Q * max_runs}")
    need = int(_lib.load(raw=True).zh_mask_runs_workspace_size(B * Q, W))
    ws = torch.empty(need, dtype=torch.uint8, device=masks_u8.device)
    _lib.check(L.zh_mask_runs_kept(_p(masks_u8), _p(kept_index), _p(kept_count), B, Q, H, W, max_runs, _p(pos), pos.numel() if packed else 0,
                                   _p(nr), _p(ba), _p(ws), need, _stream()), "zh_mask_runs_kept")


def mask_rle_kept(pos_packed, nr, kept_count, B, Q, max_runs, HW, out, out_len):
    """COCO RLE strings of the kept masks on the device from mask_runs_kept(packed=True)'s list: out u8 (any length) takes mask b*Q + j's
    string at 5 * off + 16 * rank (see include/zutis_hip.h), out_len int32 [B*Q] its length (-1: the caller encodes that mask itself)."""
    L = _lib.load()
    _chk(pos_packed, torch.int32, "mask_rle_kept positions"); _chk(nr, torch.int32, "mask_rle_kept nruns"); _chk(kept_count, torch.int32, "kept_count")
    _chk(out, torch.uint8, "mask_rle_kept out"); _chk(out_len, torch.int32, "mask_rle_kept out_len")
    _lib.check(L.zh_mask_rle_kept(_p(pos_packed), pos_packed.numel(), _p(nr), _p(kept_count), B, Q, max_runs, HW, _p(out), out.numel(), _p(out_len),
                                  _stream()), "zh_mask_rle_kept")


def mask_rle_fused_supported(H, W, max_runs) -> bool:
    return bool(_lib.load(raw=True).zh_mask_rle_fused_supported(int(H), int(W), int(max_runs)))


def mask_rle_fused_kept(masks_u8, kept_index, kept_count, max_runs, out, cursor, info, bits=None):
    """Runs, box, area and COCO RLE string of the kept masks in one launch (include/zutis_hip.h): masks u8 [B,Q,H,W], kept_index int32
    [B,Q] / kept_count int32 [B] = zh_mask_nms' outputs; out u8 (any length) takes the strings, cursor int32 [1] (zeroed by the caller)
    places them, info int32 [B*Q, 8] = (offset, length or -1, xmin, ymin, xmax, ymax, area, transitions) per kept slot.  bits: the masks
    bit-packed by mask_iou_counts(..., workspace=) — int64 [B, Q, (H*W + 63) // 64] — read instead of the bytes."""
    L = _lib.load()
    _chk(masks_u8, torch.uint8, "mask_rle_fused_kept masks"); _chk(kept_index, torch.int32, "kept_index"); _chk(kept_count, torch.int32, "kept_count")
    _chk(out, torch.uint8, "mask_rle_fused_kept out"); _chk(cursor, torch.int32, "cursor"); _chk(info, torch.int32, "info")
    B, Q, H, W = masks_u8.shape
    if bits is not None:
        _chk(bits, torch.int64, "mask_rle_fused_kept bits")
        if bits.numel() != B * Q * ((H * W + 63) // 64):
            raise ZutisHipError(f"mask_rle_fused_kept: bits holds {bits.numel()} words, {B * Q * ((H * W + 63) // 64)} expected")
    _lib.check(L.zh_mask_rle_fused_kept(_p(masks_u8), None if bits is None else _p(bits), _p(kept_index), _p(kept_count), B, Q, H, W, max_runs,
                                        _p(out), out.numel(), _p(cursor), _p(info), _stream()), "zh_mask_rle_fused_kept")


# ---------------------------------------------------------------------------------------- bilateral solver (float64)
def denormalize_u8(x, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """utils/utils.py:261-273 on device: x f32 [3,H,W] -> rgb u8 [H,W,3]."""
    import ctypes as C
    L = _lib.load()
    _chk(x, f32, "denormalize x")
    _, H, W = x.shape
    out = torch.empty((H, W, 3), dtype=torch.uint8, device=x.device)
    m = (C.c_float * 3)(*[float(np.float32(v)) for v in mean])
    s = (C.c_float * 3)(*[float(np.float32(v)) for v in std])
    _lib.check(L.zh_denormalize_u8(_p(x), _p(out), H, W, m, s, _stream()), "zh_denormalize_u8")
    return out


def bgrid_coords(rgb_u8, sigma_spatial=16, sigma_luma=16, sigma_chroma=8):
    L = _lib.load()
    H, W, _ = rgb_u8.shape
    out = torch.empty((H * W, 5), dtype=torch.int32, device=rgb_u8.device)
    _lib.check(L.zh_bgrid_coords(_p(rgb_u8), H, W, float(sigma_spatial), float(sigma_luma), float(sigma_chroma), _p(out), _stream()),
               "zh_bgrid_coords")
    return out


def bilateral_solve(rgb_u8, target, sigma_spatial=16, sigma_luma=16, sigma_chroma=8, confidence=0.999, lam=256.0,
                    a_diag_min=1e-5, cg_tol=1e-5, cg_maxiter=25, debug=False):
    """rgb u8 [H,W,3] + target u8|f64 [H,W] (device) -> soft f64 [H,W] (device), stats int32 [2] (device)
    [, n, m f64 [H*W] when debug].  A batch ([B,H,W,3] + [B,H,W]) returns [B,H,W], [B,2] (, [B,H*W] x 2): one sequence of
    launches for all B images."""
    L = _lib.load()
    batched = rgb_u8.dim() == 4
    r4 = rgb_u8 if batched else rgb_u8[None]
    t3 = target if batched else target[None]
    B, H, W, _ = r4.shape
    _chk(r4, torch.uint8, "rgb")
    assert t3.shape == (B, H, W) and t3.is_contiguous() and t3.dtype in (torch.uint8, torch.float64)
    need = B * L.zh_bilateral_workspace_size(H, W, float(sigma_spatial), float(sigma_luma), float(sigma_chroma))
    dev = r4.device
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    out = torch.empty((B, H, W), dtype=torch.float64, device=dev)
    stats = torch.zeros((B, 2), dtype=torch.int32, device=dev)
    n = torch.zeros((B, H * W), dtype=torch.float64, device=dev) if debug else None
    m = torch.zeros((B, H * W), dtype=torch.float64, device=dev) if debug else None
    t8 = t3 if t3.dtype == torch.uint8 else None
    t64 = t3 if t3.dtype == torch.float64 else None
    _lib.check(L.zh_bilateral_solve_batch(_p(r4), _p(t8), _p(t64), B, H, W, float(sigma_spatial), float(sigma_luma), float(sigma_chroma),
                                          float(confidence), float(lam), float(a_diag_min), float(cg_tol), int(cg_maxiter), _p(out),
                                          _p(stats), _p(n), _p(m), _p(ws), need, _stream()), "zh_bilateral_solve_batch")
    if not batched:
        out, stats = out[0], stats[0]
        n, m = (n[0], m[0]) if debug else (None, None)
    return (out, stats, n, m) if debug else (out, stats)


def threshold_f64_u8(x, threshold=0.5):
    """x f64 (device, contiguous) -> u8 {0,1} of the same shape: x > threshold."""
    L = _lib.load()
    _chk(x, torch.float64, "threshold x")
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    _lib.check(L.zh_threshold_f64_u8(_p(x), float(threshold), _p(out), x.numel(), _stream()), "zh_threshold_f64_u8")
    return out


def select_upsample_mask(obj, masks, out_u8, index, B, Q, h, w, H, W, scale_h, scale_w, threshold=0.5):
    L = _lib.load()
    _chk(obj, f32, "objectness"); _chk(masks, f32, "masks"); _chk(out_u8, torch.uint8, "mask out")
    _lib.check(L.zh_select_upsample_mask(_p(obj), _p(masks), _p(out_u8), _p(index), B, Q, h, w, H, W, float(scale_h), float(scale_w),
                                         float(threshold), _stream()), "zh_select_upsample_mask")


def resize_nearest_u8(x_u8, H, W):
    """F.interpolate(x[None,None], size=(H,W), mode="nearest")[0,0] for a u8 [h,w] mask on the GPU."""
    L = _lib.load()
    _chk(x_u8, torch.uint8, "resize_nearest x")
    h, w = x_u8.shape
    out = torch.empty((H, W), dtype=torch.uint8, device=x_u8.device)
    _lib.check(L.zh_resize_nearest_u8(_p(x_u8), _p(out), h, w, H, W, lin_scale(h, H), lin_scale(w, W), _stream()), "zh_resize_nearest_u8")
    return out


def topk_rows(scores, k, N=None, with_values=False, idx_map=None, idx_add=0, out_idx=None, out_val=None):
    """scores f32 [R, ld] on the GPU -> int64 [R,k] indices of the k largest of the first N columns (score desc, column asc).
    Reported index of column i = idx_map[r, i] (int64 [R, ld]) if given, else i + idx_add.  out_idx / out_val may be column
    blocks [R, k] of wider row-major tables (same row stride for both)."""
    L = _lib.load()
    _chk(scores, f32, "topk scores")
    R, ld = scores.shape
    N = ld if N is None else N
    if idx_map is not None:
        _chk(idx_map, torch.int64, "topk idx_map")
        if tuple(idx_map.shape) != (R, ld):
            raise _lib.ZutisHipError("topk_rows: idx_map must have the shape of scores")
    idx = torch.empty((R, k), dtype=torch.int64, device=scores.device) if out_idx is None else out_idx
    val = (torch.empty((R, k), dtype=f32, device=scores.device) if out_val is None else out_val) if with_values else None
    if idx.dtype != torch.int64 or idx.stride(1) != 1 or (val is not None and (val.dtype != f32 or val.stride() != idx.stride())):
        raise _lib.ZutisHipError("topk_rows: outputs must be int64 / f32 row-major blocks with equal row strides")
    _lib.check(L.zh_topk_rows(_p(scores), ld, R, N, k, _p(idx_map), int(idx_add), _p(idx), _p(val), idx.stride(0), _stream()),
               "zh_topk_rows")
    return (idx, val) if with_values else idx


def mask_runs(masks_u8, sel, max_runs=8192):
    """masks u8 [n,H,W] (device), sel int32 [m] (device) -> (positions int32 [m,max_runs], nruns int32 [m,2], box_area int32 [m,5])."""
    L = _lib.load()
    _chk(masks_u8, torch.uint8, "mask_runs masks")
    _chk(sel, torch.int32, "mask_runs sel")
    n, H, W = masks_u8.shape
    m = sel.numel()
    pos = torch.empty((m, max_runs), dtype=torch.int32, device=masks_u8.device)
    nr = torch.empty((m, 2), dtype=torch.int32, device=masks_u8.device)
    ba = torch.empty((m, 5), dtype=torch.int32, device=masks_u8.device)
    need = int(_lib.load(raw=True).zh_mask_runs_workspace_size(m, W))
    ws = torch.empty(need, dtype=torch.uint8, device=masks_u8.device)
    _lib.check(L.zh_mask_runs(_p(masks_u8), _p(sel), m, H, W, max_runs, _p(pos), _p(nr), _p(ba), _p(ws), need, _stream()), "zh_mask_runs")
    return pos, nr, ba
