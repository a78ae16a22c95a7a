"""ZUTIS forward/predict as a plan over libzutis_hip kernels (MI355X-native; no torch compute ops).

Data layout in HBM (one GPU, B images, T = 1 + h*w encoder tokens, M = 4*h*w decoder memory tokens):
  X        f32 [B*T, D]        residual stream (fp32 end to end)
  Y16      f16 [B*T, D]        LayerNorm outputs (GEMM A operands are fp16, accumulate fp32)
  QKV16    f16 [B*T, 3D]       packed q|k|v, consumed in place by flash attention (strided heads)
  H16      f16 [B*T, 4D]       QuickGELU(c_fc) — never stored in fp32
  TOK16    f16 [B*M, D]        x2-upsampled patch tokens (A operand of ffn1 and of the text-space projection)
  F2X      f16 [B*M, 320]      ffn1's second hidden layer (256) | 1 | 0...: input of the composed K / V projections and the
                               mask einsum's operand (decoder_input = ffn1's last Linear of it is never formed)
  KALL/VALL f16 [B*M, L*D]     cross-attention K / V of all L decoder layers from ONE GEMM each (K = 256: ffn1's last Linear
                               composed in at pack time; the sine-PE term enters as two small fp32 tables in the K epilogue)
  weights  f16, packed once per parameter version ([N,K] row-major = torch Linear layout, K contiguous)

Precision (DESIGN.md "Precision"): the reference computes in fp32 end to end.  Every contraction here is a *site* with a
mode: "f16" = fp16 MFMA operands, fp32 accumulate; "x3" = the reference-equivalent mode — operands carried as fp16 split
pairs (hi + lo, 22 bits) and three MFMA products per accumulator (zh_gemm_f16x3, split-pair scores in flash attention).
`precision=` picks the map: "exact" = x3 everywhere; "fast" (default) = x3 on the contractions whose rounding reaches an
output directly (ffn1, ffn2, mask einsum, text-space projection, class logits) and f16 in the transformer bodies, which
tests/test_precision_gpu.py holds to the north-star tolerance on the outlier-channel stress model; "f16" = no x3 at all.

Reference call sites are cited per step (paths relative to the reference root).
"""
from __future__ import annotations

import math
import weakref
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib, compose, ops
from ._lib import ZutisHipError

from .ops import Act


def P_shape0(w) -> int:
    """Rows (= output features) of a packed weight, plain fp16 tensor or split pair."""
    return (w.hi if isinstance(w, Act) else w).shape[0]

f16, f32 = torch.float16, torch.float32

# contraction sites (see the module docstring)
ENCODER_SITES = ("conv", "qkv", "attn", "out", "fc", "proj")
DECODER_SITES = ("dec_kv", "dec")
HEAD_SITES = ("ffn1", "ffn2", "mask", "textproj", "logits", "embed")
ALL_SITES = ENCODER_SITES + DECODER_SITES + HEAD_SITES
PRECISIONS = {
    "f16": frozenset(),
    "fast": frozenset(HEAD_SITES),
    "exact": frozenset(ALL_SITES),
}


def resolve_precision(precision) -> frozenset:
    """"f16" | "fast" | "exact" | an iterable of site names -> the set of sites computed in the x3 mode."""
    if isinstance(precision, str):
        if precision not in PRECISIONS:
            raise ZutisHipError(f"precision {precision!r} not in {sorted(PRECISIONS)}")
        return PRECISIONS[precision]
    sites = set(precision)
    bad = sites - set(ALL_SITES)
    if bad:
        raise ZutisHipError(f"unknown precision sites {sorted(bad)} (known: {ALL_SITES})")
    # a split-pair consumer needs the producer of its operand to write lo planes, and a buffer allocated as a pair must be
    # filled as one: close the set under those requirements (each rule: consumer site => the x3 GEMM that produces its operand)
    if "attn" in sites:
        sites.add("qkv")                      # Q / K / V lo planes come from the x3 QKV projection
    if "proj" in sites:
        sites.add("fc")                       # H16 (QuickGELU(c_fc)) is c_proj's operand: its lo plane comes from the x3 c_fc
    if sites & {"mask", "dec_kv"}:
        sites.add("ffn1")                     # their operand (ffn1's hidden layer, F2X) gets its lo plane from the x3 ffn1
    if "dec" in sites and "dec_kv" in sites:
        sites.add("ffn1")
    return frozenset(sites)


def _rup(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class _EngineBase:
    """Shared plumbing: fp16 weight packing keyed on parameter versions, shape-keyed buffer cache, and the two
    kernel sequences both networks share — pre-LN ViT blocks and the post-norm DETR-style decoder."""

    params: Dict[str, torch.Tensor]

    def _init_base(self, precision="exact"):
        self.x3_sites = resolve_precision(precision)
        self.precision = precision if isinstance(precision, str) else "custom"
        self._packed_key = None
        self._w: Dict[str, torch.Tensor] = {}
        self._geo: Dict[Tuple[int, int], Dict[str, torch.Tensor]] = {}
        self._bufs: Dict[Tuple, torch.Tensor] = {}
        self._buf_gen = 0             # bumped on every (re)allocation: launch plans check it
        self._buf_const: Dict[str, torch.Tensor] = {}   # buffers with constant regions: name -> the tensor that was initialised
        self._pt16_of = self._text16_of = None   # which tensors the cached f16 copies "pt16" / "text16" were made from

    def fork(self):
        """A second engine over the SAME parameters and packed weights with its own activation buffers: one per HIP stream
        when independent inputs are processed concurrently (buffers are the only mutable state of an engine)."""
        import copy
        self._pack()
        e = copy.copy(self)
        e._bufs, e._buf_gen, e._buf_const = {}, 0, {}
        # input-independent tables are shared; captured graphs are not (they replay into the parent's buffers and stream)
        e._geo = {k: v for k, v in self._geo.items() if not (isinstance(k, tuple) and k and k[0] == "graph")}
        e._pt16_of = e._text16_of = None       # provenance of the f16 copies held in the (new, empty) buffer cache
        return e

    def _version_key(self):
        return tuple((p.data_ptr(), p._version) for p in self.params.values())

    def _device(self):
        dev = next(iter(self.params.values())).device
        if dev.type != "cuda":
            raise ZutisHipError("engine parameters must live on a GPU (no CPU fallback)")
        return dev

    _GEO_CAP = 64     # per-(h,w) tables (pos-embed, sine PE, graphs): native-resolution eval sees many shapes; keep the newest

    def _geo_put(self, key, value):
        if len(self._geo) >= self._GEO_CAP:
            self._geo.pop(next(iter(self._geo)))
        self._geo[key] = value

    def _buf(self, name: str, shape, dtype) -> torch.Tensor:
        k = (name, tuple(shape), dtype)
        b = self._bufs.get(k)
        if b is None:
            for kk in [kk for kk in self._bufs if kk[0] == name]:
                del self._bufs[kk]
            b = torch.empty(shape, dtype=dtype, device=self._device())
            self._bufs[k] = b
            self._buf_gen += 1
        return b

    def _x3(self, *sites) -> bool:
        return any(s in self.x3_sites for s in sites)

    def _abuf(self, name: str, shape, split: bool) -> Act:
        """Cached fp16 activation buffer, a split pair when a consumer runs in the x3 mode."""
        return Act(self._buf(name, ((2 if split else 1),) + tuple(shape), f16))

    @staticmethod
    def _h(t):
        return t.detach().to(f16).contiguous()

    def _hw(self, t, site) -> "torch.Tensor | Act":
        """A [N,K] weight packed for its site: plain fp16, or a scaled split pair for the x3 mode."""
        return ops.split_weight(t.detach().contiguous()) if self._x3(site) else self._h(t)

    def _gemm(self, site, A, W, out, **kw):
        """One contraction at its site's precision.  x3: A and W must be split pairs (the producers were told so)."""
        if self._x3(site):
            return ops.gemm_x3(A, W, out, **kw)
        return ops.gemm(A, W, out, **kw)

    @staticmethod
    def _c32(t):
        return t.detach().to(f32).contiguous()

    def _pack_decoder(self, w, P, D, n_layers, memory_linear=None):
        """decoder.layers.{i}.* (transformer.py:231-251) -> dec.{i}.*; the cross-attention K / V weights of all layers
        are concatenated so the memory tokens are projected by ONE GEMM each.

        memory_linear = (W2 [D, F], b2 [D]): the memory is itself the output of a Linear layer, memory = f @ W2^T + b2
        (ZUTIS: the last layer of ffn1, zutis.py:500-503).  The two projections are then composed at pack time (fp64 products,
        rounded to fp32 once) so they contract over F instead of D:
            K_all = (memory + pos) @ Wk^T + bk = f @ (Wk W2)^T + (Wk b2 + bk) + pos @ Wk^T
            V_all =  memory        @ Wv^T + bv = f @ (Wv W2)^T + (Wv b2 + bv)
        and "ca_k_pos_w" keeps Wk in fp32 for the per-geometry `pos @ Wk^T` tables (ZutisEngine._geometry)."""
        c32 = self._c32
        h = lambda t: self._hw(t, "dec")
        kw, kb, vw, vb = [], [], [], []
        qpos = P["query_embed"].detach()
        for i in range(n_layers):
            p, q = f"decoder.layers.{i}.", f"dec.{i}."
            sw, sb = P[p + "self_attn.in_proj_weight"].detach(), P[p + "self_attn.in_proj_bias"].detach()
            cw, cb = P[p + "multihead_attn.in_proj_weight"].detach(), P[p + "multihead_attn.in_proj_bias"].detach()
            # q = k = tgt + query_pos, v = tgt (transformer.py:272-275) and the cross-attention query tgt + query_pos (:281-282):
            # every projection runs on tgt alone — ONE N = 3D GEMM for the self-attention's q | k | v — and starts from a
            # per-query row table that carries query_pos @ W^T + b (compose.query_pos_tables; fp64 products, stored fp32)
            w[q + "sa_qkv_w"] = h(sw)
            w[q + "sa_tab"], w[q + "ca_q_tab"] = compose.query_pos_tables(qpos, sw, sb, cw[:D], cb[:D])
            w[q + "sa_o_w"], w[q + "sa_o_b"] = h(P[p + "self_attn.out_proj.weight"]), c32(P[p + "self_attn.out_proj.bias"])
            w[q + "ca_q_w"] = h(cw[:D])
            kw.append(cw[D:2 * D]); kb.append(cb[D:2 * D]); vw.append(cw[2 * D:]); vb.append(cb[2 * D:])
            w[q + "ca_o_w"], w[q + "ca_o_b"] = h(P[p + "multihead_attn.out_proj.weight"]), c32(P[p + "multihead_attn.out_proj.bias"])
            w[q + "l1_w"], w[q + "l1_b"] = h(P[p + "linear1.weight"]), c32(P[p + "linear1.bias"])
            w[q + "l2_w"], w[q + "l2_b"] = h(P[p + "linear2.weight"]), c32(P[p + "linear2.bias"])
            for n in ("norm1", "norm2", "norm3"):
                w[q + n + ".w"], w[q + n + ".b"] = c32(P[p + n + ".weight"]), c32(P[p + n + ".bias"])
        kw, kb, vw, vb = (torch.cat(t, 0).detach() for t in (kw, kb, vw, vb))                        # [L*D, D], [L*D]
        if memory_linear is not None:
            w["ca_k_pos_w"] = c32(kw)
            kw, kb, vw, vb = compose.compose_memory_linear(kw, kb, vw, vb, *memory_linear)           # [L*D, F]
        w["ca_k_w"], w["ca_k_b"] = self._hw(kw.to(f32), "dec_kv"), c32(kb)
        w["ca_v_w"], w["ca_v_b"] = self._hw(vw.to(f32), "dec_kv"), c32(vb)
        w["dec.norm.w"], w["dec.norm.b"] = c32(P["decoder.norm.weight"]), c32(P["decoder.norm.bias"])

    def _pack_clip_visual(self, w, P, prefix: str, D: int, layers: int, patch: int):
        """CLIP VisionTransformer parameters (clip_arch.py:335-354) -> conv (K padded to 64), enc.{i}.*, ln_pre/ln_post."""
        c32 = self._c32
        kc = 3 * patch * patch
        self.Kc = _rup(kc, 64)
        wc = torch.zeros((D, self.Kc), dtype=f32, device=self._device())
        wc[:, :kc] = P[prefix + "conv1.weight"].detach().reshape(D, kc)
        w["conv"] = self._hw(wc, "conv")
        for name in ("class_embedding", "positional_embedding", "ln_pre.weight", "ln_pre.bias", "ln_post.weight", "ln_post.bias"):
            w["encoder." + name] = c32(P[prefix + name])
        self._pack_resblocks(w, P, prefix, layers)
        w["projT"] = self._hw(P[prefix + "proj"].detach().t(), self._proj_site)        # [E, D]

    _proj_site = "textproj"      # the site of the visual projection: text-space tokens (ZUTIS) / the CLS embedding (encode_image)

    def _pack_resblocks(self, w, P, prefix: str, layers: int):
        """{prefix}transformer.resblocks.{i}.* (ResidualAttentionBlock, clip_arch.py:300-321) -> enc.{i}.*"""
        hw, c32 = self._hw, self._c32
        for i in range(layers):
            p, q = f"{prefix}transformer.resblocks.{i}.", f"enc.{i}."
            w[q + "qkv_w"], w[q + "qkv_b"] = hw(P[p + "attn.in_proj_weight"], "qkv"), c32(P[p + "attn.in_proj_bias"])
            w[q + "out_w"], w[q + "out_b"] = hw(P[p + "attn.out_proj.weight"], "out"), c32(P[p + "attn.out_proj.bias"])
            w[q + "fc_w"], w[q + "fc_b"] = hw(P[p + "mlp.c_fc.weight"], "fc"), c32(P[p + "mlp.c_fc.bias"])
            w[q + "proj_w"], w[q + "proj_b"] = hw(P[p + "mlp.c_proj.weight"], "proj"), c32(P[p + "mlp.c_proj.bias"])
            for ln, ln2 in (("ln_1", "ln1"), ("ln_2", "ln2")):
                w[q + ln2 + ".w"], w[q + ln2 + ".b"] = c32(P[p + ln + ".weight"]), c32(P[p + ln + ".bias"])

    def _clip_trunk(self, x: torch.Tensor, pos: torch.Tensor, h: int, w: int):
        """conv1-as-GEMM, cls concat + pos + ln_pre, all resblocks (clip_arch.py:378-401).  Returns X f32 [B*T, D]."""
        W_, D, p = self._w, self.D, self.patch
        B = x.shape[0]
        T, R = 1 + h * w, B * (1 + h * w)
        col = self._abuf("col", (B * h * w, self.Kc), self._x3("conv"))
        ops.im2col(x, col, p, self.Kc)                                                   # :378 conv1 as GEMM
        pe32 = self._buf("patch_emb", (B * h * w, D), f32)
        self._gemm("conv", col, W_["conv"], pe32)
        X = self._buf("X", (R, D), f32)
        ops.assemble_tokens_ln(pe32, W_["encoder.class_embedding"], pos, W_["encoder.ln_pre.weight"],
                               W_["encoder.ln_pre.bias"], 1e-5, X, B, T, D)                # :384-397
        self._vit_blocks(X, B, T, D, self.heads, self.layers, 1e-5, ops.ACT_QUICKGELU)     # :318-321
        return X

    def _vit_blocks(self, X, B, T, D, heads, n_layers, eps, act, causal=False):
        """Pre-LN transformer blocks on the fp32 residual stream X [B*T, D] (in place).
        clip_arch.py:318-321 (QuickGELU, eps 1e-5) and selfmask/vision_transformer.py:160-170 (erf GELU, eps 1e-6)."""
        W_, R = self._w, B * T
        Fd = P_shape0(W_["enc.0.fc_w"])
        Y = self._abuf("Y16", (R, D), self._x3("qkv", "fc"))
        xa = self._x3("attn")                              # split-pair scores: lo planes written by the x3 QKV GEMM
        QKV = self._abuf("QKV16", (R, 3 * D), xa)
        O = self._abuf("O16", (R, D), self._x3("out"))
        Hh = self._abuf("H16", (R, Fd), self._x3("proj"))
        q_, k_, v_ = QKV, QKV.view(QKV.hi[:, D:]), QKV.view(QKV.hi[:, 2 * D:])
        for i in range(n_layers):
            pp = f"enc.{i}."
            ops.layernorm(X, W_[pp + "ln1.w"], W_[pp + "ln1.b"], eps, R, D, out_f16=Y)
            self._gemm("qkv", Y, W_[pp + "qkv_w"], QKV, bias=W_[pp + "qkv_b"])
            ops.attention(q_, k_, v_, O, batch=B, heads=heads, Tq=T, Tk=T, head_dim=D // heads,
                          ldq=3 * D, ldk=3 * D, ldv=3 * D, ldo=D, strideQ=T * 3 * D, strideK=T * 3 * D, strideV=T * 3 * D,
                          strideO=T * D, causal=causal, x3=xa)
            self._gemm("out", O, W_[pp + "out_w"], X, bias=W_[pp + "out_b"], residual=X)
            ops.layernorm(X, W_[pp + "ln2.w"], W_[pp + "ln2.b"], eps, R, D, out_f16=Y)
            self._gemm("fc", Y, W_[pp + "fc_w"], Hh, bias=W_[pp + "fc_b"], act=act)
            self._gemm("proj", Hh, W_[pp + "proj_w"], X, bias=W_[pp + "proj_b"], residual=X)

    def _decoder_kv(self, VIN16, KIN16, B, M, D, L, k_pos=None):
        """Cross-attention K / V of all L layers (transformer.py:281-284) from ONE GEMM each: [B*M, L*D] fp16 (split pairs
        when both the projection and the decoder run x3).  VIN16 / KIN16 are the value / key inputs in the layout the packed
        "ca_v_w" / "ca_k_w" contract over; k_pos = (Ty [h2, L*D], Tx [w2, L*D]): fp32 tables, K row m = y * w2 + x starts
        from Ty[y] + Tx[x] (the `pos` term of `memory + pos` when the projections were composed at pack time)."""
        W_ = self._w
        xk = self._x3("dec") and self._x3("dec_kv")                                        # K lo planes feed the x3 scores
        KALL = self._abuf("KALL", (B * M, L * D), xk)
        VALL = self._abuf("VALL", (B * M, L * D), xk)
        self._gemm("dec_kv", KIN16, W_["ca_k_w"], KALL, bias=W_["ca_k_b"], pos=k_pos)
        self._gemm("dec_kv", VIN16, W_["ca_v_w"], VALL, bias=W_["ca_v_b"])
        return KALL, VALL

    def _decoder(self, KALL, VALL, B, M, D, Q, L, heads, stack_all: bool):
        """transformer.py:114-152 over :262-291 (post-norm), tgt = zeros, query_pos = query_embed, on the projected memory
        of _decoder_kv.  Returns f16 rows with decoder.norm applied: every layer stacked as [B,L,Q,D] (stack_all) or the last
        layer only [B*Q, D]; the fp32 copy of the last layer's normed output is left in buffer "dec_out32"."""
        W_, dh, R = self._w, D // heads, B * Q
        xd = self._x3("dec")
        Ff = P_shape0(W_["dec.0.l1_w"])
        xk = bool(KALL.plane)
        tgt = self._buf("tgt", (R, D), f32)
        t1 = self._buf("t1", (R, D), f32)
        tgt16 = self._abuf("tgt16", (R, D), xd)
        qkv16 = self._abuf("dqkv16", (R, 3 * D), xd)
        qc16 = self._abuf("qc16", (R, D), xd)
        o16 = self._abuf("do16", (R, D), xd)
        ff16 = self._abuf("ff16", (R, Ff), xd)
        inter16 = self._abuf("inter16", (B * (L if stack_all else 1) * Q, D), self._x3(*self._dec_out_sites))
        out32 = self._buf("dec_out32", (R, D), f32)
        # `tgt + query_pos` never exists: the row tables of _pack_decoder enter the GEMMs as a row-periodic residual (row m gets
        # table[m % Q], added in fp32 to the finished accumulator, before the one rounding to fp16 / a split pair).  NOT as an
        # accumulator start value (the `pos` form): with large query embeddings the table dwarfs the products and every MFMA
        # then accumulates at the table's ulp — measured 0.05 on the mask proposals of the config-3 fixture (queries x20)

        def self_attention_block(pp, src16, residual, norm_out32, norm_out16):
            self._gemm("dec", src16, W_[pp + "sa_qkv_w"], qkv16, residual=W_[pp + "sa_tab"], res_rows=Q)   # q | k | v in ONE N = 3D GEMM
            ops.attention(qkv16, qkv16.view(qkv16.hi[:, D:]), qkv16.view(qkv16.hi[:, 2 * D:]), o16, batch=B, heads=heads, Tq=Q, Tk=Q,
                          head_dim=dh, ldq=3 * D, ldk=3 * D, ldv=3 * D, ldo=D, strideQ=Q * 3 * D, strideK=Q * 3 * D, strideV=Q * 3 * D,
                          strideO=Q * D, x3=xd)
            self._gemm("dec", o16, W_[pp + "sa_o_w"], t1, bias=W_[pp + "sa_o_b"], residual=residual)
            ops.layernorm(t1, W_[pp + "norm1.w"], W_[pp + "norm1.b"], 1e-5, R, D, out_f32=norm_out32, out_f16=norm_out16)
        # tgt = zeros (zutis.py:164) and query_pos is a parameter, so layer 0's whole self-attention block — projections of
        # (0 + query_pos, 0), attention over the Q queries, out-projection, norm1 — does not depend on the image: its result
        # (tgt after norm1, fp32 and fp16; [R, D] = the same Q rows for every image) is computed once per (batch rows, parameter
        # version) with the same kernels and cached; layer 0 then starts at the cross-attention.
        ikey = ("dec_init", R, self._packed_key)
        init = self._geo.get(ikey)
        if init is None:
            dev = self._device()
            z16 = Act(torch.zeros((2 if xd else 1, R, D), dtype=f16, device=dev))
            init = {"tgt0": torch.empty((R, D), dtype=f32, device=dev), "tgt0_16": Act.empty((R, D), xd, dev)}
            self_attention_block("dec.0.", z16, None, init["tgt0"], init["tgt0_16"])      # q = k = query_pos, v = 0, + tgt (= 0)
            self._geo_put(ikey, init)
        for l in range(L):
            pp = f"dec.{l}."
            if l == 0:
                tgt_in, tgt_in16 = init["tgt0"], init["tgt0_16"]
            else:
                self_attention_block(pp, tgt16, tgt, tgt, tgt16)                            # transformer.py:272-278
                tgt_in, tgt_in16 = tgt, tgt16
            self._gemm("dec", tgt_in16, W_[pp + "ca_q_w"], qc16, residual=W_[pp + "ca_q_tab"], res_rows=Q)   # :281-282 query projection
            ops.attention(qc16, KALL.view(KALL.hi[:, l * D:]), VALL.view(VALL.hi[:, l * D:]), o16, batch=B, heads=heads, Tq=Q, Tk=M,
                          head_dim=dh, ldq=D, ldk=L * D, ldv=L * D, ldo=D, strideQ=Q * D, strideK=M * L * D, strideV=M * L * D,
                          strideO=Q * D, x3=xk)
            self._gemm("dec", o16, W_[pp + "ca_o_w"], t1, bias=W_[pp + "ca_o_b"], residual=tgt_in)
            ops.layernorm(t1, W_[pp + "norm2.w"], W_[pp + "norm2.b"], 1e-5, R, D, out_f32=tgt, out_f16=tgt16)
            self._gemm("dec", tgt16, W_[pp + "l1_w"], ff16, bias=W_[pp + "l1_b"], act=ops.ACT_RELU)
            self._gemm("dec", ff16, W_[pp + "l2_w"], t1, bias=W_[pp + "l2_b"], residual=tgt)
            ops.layernorm(t1, W_[pp + "norm3.w"], W_[pp + "norm3.b"], 1e-5, R, D, out_f32=tgt, out_f16=tgt16)
            if stack_all:                                                                   # :140-150, stacked [B,L,Q,D]
                ops.layernorm(tgt, W_["dec.norm.w"], W_["dec.norm.b"], 1e-5, R, D, out_f16=inter16,
                              out_group_rows=Q, out_group_stride=L * Q, out_offset=l * Q)
            elif l == L - 1:
                ops.layernorm(tgt, W_["dec.norm.w"], W_["dec.norm.b"], 1e-5, R, D, out_f16=inter16, out_f32=out32)
        return inter16

    _dec_out_sites = ("ffn2",)   # sites consuming the decoder's normed outputs (ZUTIS: ffn2; SelfMask: mask einsum + objectness MLP)


class ZutisEngine(_EngineBase):
    """Inference engine for one ZUTIS network.  `params` maps reference state_dict keys to fp32 CUDA tensors
    (typically the nn.Parameters of the drop-in module, so load_state_dict() is picked up via version counters)."""

    def __init__(self, params: Dict[str, torch.Tensor], patch: int, dec_heads: int = 8, precision="exact"):
        self.params = params
        self.patch = patch
        self.D = params["encoder.class_embedding"].shape[0]
        self.heads = self.D // 64                                              # clip_arch.py:606
        self.layers = 1 + max(int(k.split(".")[3]) for k in params if k.startswith("encoder.transformer.resblocks."))
        self.dec_layers = 1 + max(int(k.split(".")[2]) for k in params if k.startswith("decoder.layers."))
        self.dec_heads = dec_heads
        self.dec_dh = self.D // dec_heads
        self.Q = params["query_embed"].shape[0]
        self.E = params["encoder.proj"].shape[1]
        self.grid = int(math.isqrt(params["encoder.positional_embedding"].shape[0] - 1))
        if self.dec_dh not in (64, 96):
            raise ZutisHipError(f"decoder head_dim {self.dec_dh} unsupported by zh_attention_f16 (64 or 96)")
        self._init_base(precision)

    # ------------------------------------------------------------------ packing
    def _pack(self):
        key = self._version_key()
        if key == self._packed_key:
            return
        P, D, w = self.params, self.D, {}
        c32 = self._c32
        self._pack_clip_visual(w, P, "encoder.", D, self.layers, self.patch)
        for ffn in ("ffn1", "ffn2"):
            for j in range(3):
                if (ffn, j) == ("ffn1", 2):
                    continue                              # composed into its consumers below: decoder_input is never formed
                w[f"{ffn}.{j}.w"], w[f"{ffn}.{j}.b"] = self._hw(P[f"{ffn}.layers.{j}.weight"], ffn), c32(P[f"{ffn}.layers.{j}.bias"])
        # decoder_input = f @ W2^T + b2 (ffn1's last Linear, zutis.py:500-503; f = its 256-wide hidden layer) has two consumers,
        # both linear in it: the decoder's K / V projections (_pack_decoder composes W2 into them) and the mask einsum
        # (zutis.py:196-198)  q . decoder_input[m] = (W2^T q) . f[m] + q . b2.  With f stored with a constant ones column
        # (FX columns: f | 1 | 0...), the einsum contracts [W2^T q | q.b2 | 0] with it over FX = 320 instead of D = 768, and
        # the [B*M, 768] decoder_input tensor and its GEMM disappear.  "mask_q.w" maps a query to that FX-vector.
        W2, b2 = P["ffn1.layers.2.weight"].detach(), P["ffn1.layers.2.bias"].detach()
        self.Fh = W2.shape[1]
        self.FX = _rup(self.Fh + 1, 64)
        wq = compose.mask_query_weight(W2, b2, self.FX)
        w["mask_q.w"] = self._hw(wq, "mask")
        self._pack_decoder(w, P, D, self.dec_layers, memory_linear=(P["ffn1.layers.2.weight"], P["ffn1.layers.2.bias"]))
        self._w, self._packed_key = w, key
        self._geo.clear()

    def _geometry(self, h: int, w: int):
        """Input-independent tables per token grid: bicubic pos-embed (clip_arch.py:356-374) and sine PE
        (positional_embedding.py:29-52) — computed once on device, cached."""
        g = self._geo.get((h, w))
        if g is None:
            D, dev = self.D, self._device()
            pos = torch.empty((1 + h * w, D), dtype=f32, device=dev)
            sh = np.float32(1.0 / ((h + 0.1) / self.grid))
            sw = np.float32(1.0 / ((w + 0.1) / self.grid))
            ops.posembed_bicubic(self._w["encoder.positional_embedding"], pos, self.grid, h, w, D, sh, sw, True)
            pe = torch.empty((4 * h * w, D), dtype=f32, device=dev)
            ops.sine_pe(pe, 2 * h, 2 * w, D)
            # `pos @ Wk^T` (transformer.py:281 through :283's key projection, all layers): the sine PE is [py(y) | px(x)]
            # (positional_embedding.py:47-52), so the term is Ty[y] + Tx[x] with two small tables (fp64 products, stored fp32)
            # that the K GEMM's accumulators start from
            tdt = f32 if self._x3("dec_kv") else f16      # fp16 K: fp16 tables (the tile's slice is staged through LDS)
            Ty, Tx = compose.separable_pos_tables(pe, self._w["ca_k_pos_w"], 2 * h, 2 * w, tdt)   # [2h, L*D], [2w, L*D]
            g = {"pos": pos, "pe": pe, "k_pos": (Ty, Tx)}
            self._geo_put((h, w), g)
        return g

    # ------------------------------------------------------------------ encoder
    def encode(self, x: torch.Tensor):
        """clip_arch.py:377-411 -> (patch tokens f32 [B,hw,D] (ln_post applied, cls dropped), h, w)."""
        tok, _, h, w = self._encode(x, False)
        return tok, h, w

    def _encode(self, x: torch.Tensor, want16: bool, want32: bool = True):
        """encode() plus, on request, the fp16 / split-pair copy of the tokens the head's GEMMs read (same LayerNorm launch)."""
        self._pack()
        if not (x.is_cuda and x.dtype == f32 and x.dim() == 4 and x.shape[1] == 3):
            raise ZutisHipError("encode: expected float32 CUDA tensor [B,3,H,W]")
        x = x.contiguous()
        W_, D, p = self._w, self.D, self.patch
        B, _, H, Wd = x.shape
        h, w = (H - p) // p + 1, (Wd - p) // p + 1
        T = 1 + h * w
        X = self._clip_trunk(x, self._geometry(h, w)["pos"], h, w)
        tok = self._buf("tok", (B, h * w, D), f32) if want32 else None
        tok16 = self._abuf("tok16", (B * h * w, D), self._x3("ffn1", "textproj")) if want16 else None
        ops.layernorm(X, W_["encoder.ln_post.weight"], W_["encoder.ln_post.bias"], 1e-5, B * h * w, D, out_f32=tok, out_f16=tok16,
                      in_group_rows=h * w, in_group_stride=T, in_offset=1)                 # :403-404
        return tok, tok16, h, w

    # ------------------------------------------------------------------ full forward
    def forward(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        """networks/zutis.py:472-532."""
        _, tok16, h, w = self._encode(x, True, want32=False)
        W_, D, B, Q, L = self._w, self.D, x.shape[0], self.Q, self.dec_layers
        h2, w2 = 2 * h, 2 * w
        M = h2 * w2
        geo = self._geometry(h, w)
        # zutis.py:491-503 upsamples the tokens x2 and then applies ffn1; zutis.py:319 projects the upsampled tokens.  Both first
        # steps are LINEAR maps of the tokens and bilinear interpolation is a convex combination (weights 0.25 / 0.75, sum 1),
        # so W.up(t) + b == up(W.t + b): the first ffn1 layer and the text-space projection run on the h*w tokens (4x fewer rows)
        # and their outputs are upsampled (ReLU after the interpolation, where the reference has it).  Same function, different
        # rounding order (fp32-class in the x3 mode); the [B, 4hw, 768] upsampled token tensor is never formed.
        Fh, FX = self.Fh, self.FX
        h1 = self._buf("ffn_h1_lo", (B * h * w, Fh), f32)
        self._gemm("ffn1", tok16, W_["ffn1.0.w"], h1, bias=W_["ffn1.0.b"])                  # :500-503 (layer 0, pre-ReLU)
        f1 = self._abuf("ffn_h1", (B * M, Fh), self._x3("ffn1"))
        ops.upsample2x_cl(h1, B, h, w, Fh, out_f16=f1, relu=True)                           # :491-495 + ReLU
        # ffn1's hidden layer 2 with the constant columns [1, 0, ...] behind it (see _pack): rows are FX wide
        f2x = self._abuf("ffn_h2x", (B * M, FX), self._x3("ffn1"))
        if self._buf_const.get("ffn_h2x") is not f2x.t:              # once per (re)allocation of the cached buffer
            f2x.t[:, :, Fh:] = 0
            f2x.t[0, :, Fh] = 1
            self._buf_const["ffn_h2x"] = f2x.t
        f2 = f2x.view(f2x.hi[:, :Fh])
        self._gemm("ffn1", f1, W_["ffn1.1.w"], f2, bias=W_["ffn1.1.b"], act=ops.ACT_RELU)
        # ffn1's last Linear (-> decoder_input) is composed into its consumers at pack time.  The decoder's K / V projections
        # of decoder_input (+ pos) contract over the hidden width (256) instead of D (768) — 3x fewer flops on what was 22 % of
        # the model's GEMM work — and `memory + pos` (transformer.py:281) is never materialised
        KALL, VALL = self._decoder_kv(f2, f2, B, M, D, L, k_pos=geo["k_pos"])
        inter16 = self._decoder(KALL, VALL, B, M, D, Q, L, self.dec_heads, stack_all=True)  # transformer.py:114-152
        RQ = B * L * Q
        g1 = self._abuf("ffn2_h1", (RQ, Fh), self._x3("ffn2"))
        g2 = self._abuf("ffn2_h2", (RQ, Fh), self._x3("ffn2"))
        q32 = self._buf("q32", (RQ, D), f32)
        q16 = self._abuf("q16", (RQ, D), self._x3("mask"))
        self._gemm("ffn2", inter16, W_["ffn2.0.w"], g1, bias=W_["ffn2.0.b"], act=ops.ACT_RELU)        # zutis.py:514
        self._gemm("ffn2", g1, W_["ffn2.1.w"], g2, bias=W_["ffn2.1.b"], act=ops.ACT_RELU)
        self._gemm("ffn2", g2, W_["ffn2.2.w"], q32, bias=W_["ffn2.2.b"])
        ops.l2norm_rows(q32, RQ, D, out_f16=q16)                                            # :515
        masks = torch.empty((B, L, Q, h2, w2), dtype=f32, device=x.device)
        qw = self._abuf("mask_q", (RQ, FX), self._x3("mask"))
        self._gemm("mask", q16, W_["mask_q.w"], qw)                                         # [W2^T q | q.b2 | 0]
        self._gemm("mask", qw, f2x, masks, act=ops.ACT_SIGMOID, M=L * Q, N=M, K=FX, lda=FX, ldw=FX, ldc=M,
                   batch=B, strideA=L * Q * FX, strideW=M * FX, strideC=L * Q * M)          # :196-198,209
        tsl = self._buf("textspace_lo", (B * h * w, self.E), f32)
        self._gemm("textproj", tok16, W_["projT"], tsl)                                     # :319 on the h*w tokens
        ts = self._buf("textspace", (B * M, self.E), f32)
        ops.upsample2x_cl(tsl, B, h, w, self.E, out_f32=ts)                                 # :491-495
        pt = torch.empty((B, h2, w2, self.E), dtype=f32, device=x.device)
        ws = self._buf("gln_ws", (max(1, ops.global_ln_l2_workspace_size(B, M, self.E)),), torch.uint8)
        pt16 = self._abuf("pt16", (B * M, self.E), self._x3("logits"))   # the copy predict_semantic's class-logit GEMM consumes
        ops.global_ln_l2(ts, B, M, self.E, out_f32=pt, out_f16=pt16, eps=1e-5, l2_eps=1e-7, workspace=ws)  # :320-322
        self._pt16_of = (weakref.ref(pt), pt._version)           # identity, not address: a freed tensor's address can be reused
        return {"mask_proposals": masks, "patch_tokens": pt}

    # ------------------------------------------------------------------ hipGraph replay (latency path)
    def forward_graphed(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Same result as forward(), replayed from a hipGraph captured once per input shape.  At batch 1-4 the eager path is
        host-bound (~230 launches x ~11 us of Python/ctypes = 2.7 ms per forward regardless of B); COCO-20K evaluation
        (coco20k_eval.py:241-268) runs batch 1.  Outputs are fresh tensors (copied out of the graph's static buffers)."""
        self._pack()
        key = ("graph", tuple(x.shape))
        g = self._geo.get(key)
        if g is not None and g["weights"] != self._packed_key:
            g = None                                 # parameters changed since capture: the graph holds the old packed weights
        if g is None:
            static_x = x.clone()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):              # warm-up on a side stream: fills the buffer / geometry caches
                for _ in range(2):
                    self.forward(static_x)
            torch.cuda.current_stream().wait_stream(s)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self.forward(static_x)
            # The graph bakes in raw pointers to this shape's scratch buffers, geometry tables and packed weights.  _buf()
            # drops a name's buffers when another shape arrives and _geo is a bounded cache, so the graph keeps its own
            # references: shape A, then B, then A again replays A into memory that is still A's.
            g = {"x": static_x, "graph": graph, "out": out, "weights": self._packed_key,
                 "keep": (dict(self._bufs), {k: v for k, v in self._geo.items() if not (isinstance(k, tuple) and k and k[0] == "graph")},
                          self._w)}
            self._geo_put(key, g)
        g["x"].copy_(x)
        g["graph"].replay()
        return {k: v.clone() for k, v in g["out"].items()}

    # ------------------------------------------------------------------ native launch plans
    def build_plan(self, x_shape, text: Optional[torch.Tensor] = None, size: Optional[Tuple[int, int]] = None):
        """Record forward() (and predict_semantic() when `text` is given) for one input shape into a native launch plan
        (zutis_amd/plan.py).  Returns a dict with the static input `x`, the static outputs and the plan.  The engine must
        not be used with other shapes between build and replay (its buffer cache backs the recorded pointers)."""
        from . import plan as zplan
        self._pack()
        dev = self._device()
        static_x = torch.zeros(tuple(x_shape), dtype=f32, device=dev)
        text32 = None if text is None else text.detach().to(device=dev, dtype=f32).contiguous()
        self.forward(static_x)                                     # eager warm-up: packs weights, fills caches
        if text32 is not None:
            self.predict_semantic(self.forward(static_x)["patch_tokens"], text32, size)
        gen = self._buf_gen
        with zplan.Recorder() as rec:
            out = self.forward(static_x)
            labels = None if text32 is None else self.predict_semantic(out["patch_tokens"], text32, size)
        if self._buf_gen != gen:
            raise ZutisHipError("build_plan: buffers were re-allocated while recording")
        return {"x": static_x, "out": out, "labels": labels, "text": text32, "plan": rec.build(), "gen": gen,
                "weights": self._packed_key}

    def run_plan(self, p, x: Optional[torch.Tensor] = None):
        """Replay a plan from build_plan() on the current stream.  Outputs are the plan's static tensors (overwritten by
        the next replay: clone what must survive)."""
        if self._buf_gen != p["gen"]:
            raise ZutisHipError("run_plan: the engine's buffers changed since the plan was built")
        if self._version_key() != p["weights"]:
            raise ZutisHipError("run_plan: parameters changed since the plan was built (it holds the old packed weights): rebuild it")
        if x is not None:
            p["x"].copy_(x)
        p["plan"].run(torch.cuda.current_stream().cuda_stream)
        return p["out"], p["labels"]

    # ------------------------------------------------------------------ predict (semantic)
    def semantic_logits_lowres(self, patch_tokens: torch.Tensor, text: torch.Tensor) -> torch.Tensor:
        """einsum("nc,bchw->bnhw") zutis.py:361-365 -> f32 [B,n,h,w]."""
        B, h, w, E = patch_tokens.shape
        n = text.shape[0]
        xl = self._x3("logits")
        pt16 = self._abuf("pt16", (B * h * w, E), xl)
        src = self._pt16_of
        if not (src is not None and src[0]() is patch_tokens and src[1] == patch_tokens._version):
            ops.cast_f16(patch_tokens.contiguous(), pt16, B * h * w, E)     # tokens not produced by the last forward()
            self._pt16_of = None
        t32 = text.detach().to(device=patch_tokens.device, dtype=f32).contiguous()
        t16 = self._abuf("text16", (n, E), xl)
        recording = _lib.RECORDER is not None                              # a launch plan always contains the cast
        src = self._text16_of
        same = src is not None and src[0]() is text and src[1] == text._version and src[2] == self._buf_gen
        if recording or not same:                                            # eager: the category embeddings rarely change
            ops.cast_f16(t32, t16, n, E)
            self._text16_of = None if recording else (weakref.ref(text), text._version, self._buf_gen)
        lo = torch.empty((B, n, h, w), dtype=f32, device=patch_tokens.device)
        self._gemm("logits", t16, pt16, lo, M=n, N=h * w, K=E, lda=E, ldw=E, ldc=h * w, batch=B, strideA=0, strideW=h * w * E,
                   strideC=n * h * w)
        return lo

    def predict_semantic(self, patch_tokens: torch.Tensor, text: torch.Tensor, size: Optional[Tuple[int, int]],
                         return_logits: bool = False):
        """zutis.py:355-372.  Labels come from the fused upsample+argmax kernel; [B,n,H,W] is never materialised."""
        lo = self.semantic_logits_lowres(patch_tokens, text)
        B, n, h, w = lo.shape
        if return_logits:
            if size is None:
                return lo
            out = torch.empty((B, n, size[0], size[1]), dtype=f32, device=lo.device)
            ops.upsample_bilinear_nchw(lo, B * n, h, w, size[0], size[1], out=out)
            return out
        H, Wd = (h, w) if size is None else (int(size[0]), int(size[1]))
        labels = torch.empty((B, H, Wd), dtype=torch.int64, device=lo.device)
        ops.upsample_argmax(lo, labels, B, n, h, w, H, Wd)
        return labels

    # ------------------------------------------------------------------ predict (instance)
    def instance_candidates(self, mask_proposals_last: torch.Tensor, patch_tokens: torch.Tensor, text: torch.Tensor,
                            threshold: float = 0.5, temperature: float = 5.0, size: Optional[Tuple[int, int]] = None):
        """zutis.py:376-423 on device: binary masks, sizes, confidence, masked-mean tokens, class + score, and the
        full-resolution thresholded masks.  Returns (masks u8 [B,Q,H,W], scores f32 [B,Q], category int64 [B,Q])."""
        self._pack()
        mp = mask_proposals_last.contiguous()
        B, Q, h, w = mp.shape
        M, E, dev = h * w, patch_tokens.shape[-1], mp.device
        pt = patch_tokens.contiguous()
        t32 = text.detach().to(device=dev, dtype=f32).contiguous()
        sizes = torch.empty((B * Q,), dtype=f32, device=dev)
        conf = torch.empty((B * Q,), dtype=f32, device=dev)
        binary = torch.empty((B, Q, h, w), dtype=torch.uint8, device=dev)
        ops.instance_mask_stats(mp, Q * M, threshold, B, Q, M, sizes, conf, binary)
        avg = torch.empty((B * Q, E), dtype=f32, device=dev)
        ops.masked_mean_tokens(pt, binary, sizes, avg, B, Q, M, E)
        cat = torch.empty((B, Q), dtype=torch.int64, device=dev)
        score = torch.empty((B, Q), dtype=f32, device=dev)
        ops.instance_classify(avg, t32, conf, temperature, B * Q, t32.shape[0], E, cat, score)
        if size is not None:
            masks = torch.empty((B, Q, size[0], size[1]), dtype=torch.uint8, device=dev)
            ops.upsample_bilinear_nchw(mp, B * Q, h, w, size[0], size[1], mask_u8=masks, threshold=threshold)
        else:
            masks = binary
        return masks, score, cat

    def mask_iou_matrix(self, masks_u8: torch.Tensor, return_areas: bool = False):
        """Pairwise IoU of one image's [Q,H,W] u8 masks: exact popcounts on device, float64 divide on the host
        (= utils/iou.py:30-32 on boolean masks).  The diagonal of the intersection counts is each mask's area."""
        n = masks_u8.shape[0]
        px = masks_u8[0].numel()
        inter = torch.empty((n, n), dtype=torch.int32, device=masks_u8.device)
        uni = torch.empty((n, n), dtype=torch.int32, device=masks_u8.device)
        ops.mask_iou_counts(masks_u8.contiguous(), n, px, inter, uni)
        ih = inter.cpu().numpy()
        iou = ih / (uni.cpu().numpy() + 1e-7)
        return (iou, np.diag(ih).copy()) if return_areas else iou

    def instance_nms(self, masks_u8: torch.Tensor, scores: torch.Tensor, category_ids: torch.Tensor, nms_type: str = "hard",
                     nms_threshold: float = 0.3, sigma: float = 0.5, threshold: float = 0.001):
        """zutis.py:211-299 for a batch, entirely on the device: masks u8 [B,Q,H,W], scores f32 [B,Q], category_ids int64 [B,Q]
        -> list of (batch index, category, query index, score) in the reference's emission order.  Popcount IoU counts per
        image (zh_mask_iou_counts), then one launch of the greedy per-category loop (zh_mask_nms, one workgroup per image);
        only the kept (index, score, category) triples and their count cross PCIe."""
        B, Q, H, W = masks_u8.shape
        dev = masks_u8.device
        inter = torch.empty((B, Q, Q), dtype=torch.int32, device=dev)
        uni = torch.empty((B, Q, Q), dtype=torch.int32, device=dev)
        m = masks_u8.contiguous()
        for b in range(B):
            ops.mask_iou_counts(m[b], Q, H * W, inter[b], uni[b])
        idx, sc, cat, cnt = ops.mask_nms(inter, uni, scores.contiguous(), category_ids.contiguous(), nms_type, nms_threshold, sigma,
                                         threshold)
        cnt_h, idx_h, sc_h, cat_h = cnt.cpu().numpy(), idx.cpu().numpy(), sc.cpu().numpy(), cat.cpu().numpy()
        # The kernel walks the categories in ascending id; the reference walks `set(category_ids_per_image)` (zutis.py:237-238), i.e.
        # CPython's iteration order of a set of numpy int64 scalars — ascending only while every id is below the hash table's
        # size.  Re-create that very set on the host (Q ids per image) and order the per-category groups by it (stable: the
        # selection order inside a category is the kernel's, which is the reference's).
        all_cat = category_ids.cpu().numpy()
        out = []
        for b in range(B):
            rank = {int(c): i for i, c in enumerate(set(all_cat[b]))}
            rows = [(b, int(cat_h[b, j]), int(idx_h[b, j]), float(sc_h[b, j])) for j in range(int(cnt_h[b]))]
            rows.sort(key=lambda r: rank[r[1]])
            out += rows
        return out

    def encode_masks(self, masks_u8: torch.Tensor, sel: np.ndarray, max_runs: int = 8192):
        """COCO RLE dicts, xyxy boxes and areas of the masks `sel` (flat indices into [n,H,W]) without moving the masks
        to the host: zh_mask_runs extracts the column-major run boundaries on the device; only those cross PCIe."""
        from . import rle
        n, H, W = masks_u8.shape
        if len(sel) == 0:
            return [], [], []
        sel_dev = torch.from_numpy(np.ascontiguousarray(sel, dtype=np.int32)).to(masks_u8.device)
        pos, nr, ba = ops.mask_runs(masks_u8.contiguous(), sel_dev, max_runs)
        nr_h, ba_h = nr.cpu().numpy(), ba.cpu().numpy()
        keep = int(min(max_runs, max(1, nr_h[:, 0].max())))
        pos_h = pos[:, :keep].cpu().numpy()
        rles, boxes, areas = [], [], []
        for j, q in enumerate(sel):
            cnt, first = int(nr_h[j, 0]), int(nr_h[j, 1])
            if cnt > max_runs:                                   # pathological mask: fall back to the host encoder
                rles.append(rle.encode(masks_u8[int(q)].cpu().numpy()))
            else:
                rles.append(rle.rle_from_transitions(pos_h[j, :cnt], first, H, W))
            boxes.append([float(v) for v in ba_h[j, :4]])
            areas.append(int(ba_h[j, 4]))
        return rles, boxes, areas


class ClipImageEncoder(_EngineBase):
    """CLIP `encode_image` for the index-dataset pipeline (utils/extract_image_embeddings.py:72-73; third-party `clip`,
    restated from the original forward kept in clip_arch.py:413-431,531-532): fixed positional embedding, CLS token ->
    ln_post -> @proj, then L2 normalisation.  `params` uses the CLIP visual state_dict keys under `prefix`."""

    _proj_site = "embed"

    def __init__(self, params: Dict[str, torch.Tensor], patch: int, prefix: str = "visual.", precision="exact"):
        self.params, self.patch, self.prefix = params, patch, prefix
        self.D = params[prefix + "class_embedding"].shape[0]
        self.heads = self.D // 64
        self.layers = 1 + max(int(k[len(prefix):].split(".")[2]) for k in params if k.startswith(prefix + "transformer.resblocks."))
        self.E = params[prefix + "proj"].shape[1]
        self.grid = int(math.isqrt(params[prefix + "positional_embedding"].shape[0] - 1))
        self._init_base(precision)

    def _pack(self):
        key = self._version_key()
        if key == self._packed_key:
            return
        w = {}
        self._pack_clip_visual(w, self.params, self.prefix, self.D, self.layers, self.patch)
        self._w, self._packed_key = w, key

    def encode_image(self, x: torch.Tensor) -> torch.Tensor:
        """x f32 [B,3,R,R] at the model's native resolution -> unit-norm embeddings f32 [B,E]."""
        self._pack()
        if not (x.is_cuda and x.dtype == f32 and x.dim() == 4 and x.shape[1] == 3):
            raise ZutisHipError("encode_image: expected float32 CUDA tensor [B,3,H,W]")
        p, g = self.patch, self.grid
        B, _, H, Wd = x.shape
        h, w = (H - p) // p + 1, (Wd - p) // p + 1
        if (h, w) != (g, g):
            raise ZutisHipError(f"encode_image: input {H}x{Wd} gives a {h}x{w} grid; CLIP's fixed pos-embed needs {g}x{g}")
        W_, D = self._w, self.D
        X = self._clip_trunk(x.contiguous(), W_["encoder.positional_embedding"], h, w)
        cls16 = self._abuf("cls16", (B, D), self._x3("embed"))
        ops.layernorm(X, W_["encoder.ln_post.weight"], W_["encoder.ln_post.bias"], 1e-5, B, D, out_f16=cls16,
                      in_group_rows=1, in_group_stride=1 + h * w, in_offset=0)              # ln_post(x[:, 0, :])
        e32 = self._buf("emb32", (B, self.E), f32)
        self._gemm("embed", cls16, W_["projT"], e32)                                        # @ proj
        out = torch.empty((B, self.E), dtype=f32, device=x.device)
        ops.l2norm_rows(e32, B, self.E, out_f32=out)                                        # / norm(dim=-1)
        return out


class ClipTextEncoder(_EngineBase):
    """CLIP text tower: `encode_text` (networks/clip_arch.py:534-547) and the prompt ensembling of
    utils/extract_text_embeddings.py:98-115.  `params` uses the CLIP state_dict keys under `prefix`
    (token_embedding.weight, positional_embedding, transformer.resblocks.*, ln_final.*, text_projection);
    heads = width // 64 (clip_arch.py:606)."""

    def __init__(self, params: Dict[str, torch.Tensor], prefix: str = "", chunk: int = 4096, precision="exact"):
        self.params, self.prefix, self.chunk = params, prefix, chunk
        self.ctx, self.D = params[prefix + "positional_embedding"].shape
        self.vocab = params[prefix + "token_embedding.weight"].shape[0]
        self.E = params[prefix + "text_projection"].shape[1]
        self.heads = self.D // 64
        k0 = len((prefix + "transformer.resblocks.").split(".")) - 1
        self.layers = 1 + max(int(k.split(".")[k0]) for k in params if k.startswith(prefix + "transformer.resblocks."))
        if self.D % 64 or self.E % 4:
            raise ZutisHipError("ClipTextEncoder: width must be a multiple of 64 and the embedding of 4")
        self._init_base(precision)

    def _pack(self):
        key = self._version_key()
        if key == self._packed_key:
            return
        P, pre, w = self.params, self.prefix, {}
        self._pack_resblocks(w, P, pre, self.layers)
        w["table"] = self._c32(P[pre + "token_embedding.weight"])
        w["pos"] = self._c32(P[pre + "positional_embedding"])
        w["lnf.w"], w["lnf.b"] = self._c32(P[pre + "ln_final.weight"]), self._c32(P[pre + "ln_final.bias"])
        w["projT"] = self._hw(P[pre + "text_projection"].detach().t(), "embed")        # [E, D]
        self._w, self._packed_key = w, key

    def _encode_chunk(self, tok: torch.Tensor, out: torch.Tensor):
        W_, D, ctx = self._w, self.D, self.ctx
        n = tok.shape[0]
        X = self._buf("X", (n * ctx, D), f32)
        ops.embed_tokens(tok, W_["table"], W_["pos"], X)                               # :535-537
        self._vit_blocks(X, n, ctx, D, self.heads, self.layers, 1e-5, ops.ACT_QUICKGELU, causal=True)   # :538-540
        eot = self._buf("eot", (n, D), f32)
        ops.eot_rows(tok, X, eot)                                                      # :545 (LN is row-wise: gather first)
        e16 = self._abuf("eot16", (n, D), self._x3("embed"))
        ops.layernorm(eot, W_["lnf.w"], W_["lnf.b"], 1e-5, n, D, out_f16=e16)          # :541 ln_final
        self._gemm("embed", e16, W_["projT"], out)                                     # @ text_projection

    def encode_text(self, tokens: torch.Tensor) -> torch.Tensor:
        """tokens int64 [n, ctx] (clip.tokenize layout: EOT = the largest id of each row) -> f32 [n, E], not normalised."""
        self._pack()
        dev = self._device()
        if tokens.dim() != 2 or tokens.shape[1] != self.ctx:
            raise ZutisHipError(f"encode_text: expected tokens [n, {self.ctx}]")
        tok = tokens.to(device=dev, dtype=torch.int64).contiguous()
        if tok.numel() and (int(tok.min()) < 0 or int(tok.max()) >= self.vocab):
            raise IndexError("encode_text: token id out of range")                      # nn.Embedding raises likewise
        n = tok.shape[0]
        out = torch.empty((n, self.E), dtype=f32, device=dev)
        for i in range(0, n, self.chunk):
            self._encode_chunk(tok[i:i + self.chunk], out[i:i + self.chunk])
        return out

    def prompt_ensemble(self, tokens: torch.Tensor) -> torch.Tensor:
        """tokens int64 [C, T, ctx] (T prompts per category) -> unit-norm f32 [C, E]: encode, L2-normalise every prompt,
        average over T, L2-normalise (extract_text_embeddings.py:104-113) — all categories in one batch, on device."""
        C, T, ctx = tokens.shape
        e = self.encode_text(tokens.reshape(C * T, ctx))
        if T == 1:
            return e                                                                    # :107-108: single template -> raw embedding
        ops.l2norm_rows(e, C * T, self.E, out_f32=e)
        out = torch.empty((C, self.E), dtype=f32, device=e.device)
        ops.group_mean_l2norm(e, out, C, T, self.E)
        return out


class SelfMaskEngine(_EngineBase):
    """SelfMask pseudo-labeller (networks/selfmask/selfmask.py:137-245): DINO ViT-S/8 encoder
    (vision_transformer.py:260-304) -> 6-layer decoder, 20 queries, no memory pos -> x2 upsampled tokens . queries ->
    objectness MLP; inference picks the argmax-objectness query, x4 bilinear, crop, > 0.5."""

    _dec_out_sites = ("mask", "ffn2")

    def __init__(self, params: Dict[str, torch.Tensor], patch: int = 8, heads: int = 6, precision="exact"):
        self.params = params
        self.patch, self.heads = patch, heads
        self.D = params["encoder.cls_token"].shape[-1]
        self.layers = 1 + max(int(k.split(".")[2]) for k in params if k.startswith("encoder.blocks."))
        self.dec_layers = 1 + max(int(k.split(".")[2]) for k in params if k.startswith("decoder.layers."))
        self.Q = params["query_embed"].shape[0]
        self.n_pos = params["encoder.pos_embed"].shape[1] - 1
        self.grid = int(math.isqrt(self.n_pos))
        if self.D // heads != 64:
            raise ZutisHipError("SelfMaskEngine: head_dim must be 64")
        self._init_base(precision)

    def _pack(self):
        key = self._version_key()
        if key == self._packed_key:
            return
        P, D, w = self.params, self.D, {}
        dev = self._device()
        hw, c32 = self._hw, self._c32
        kc = 3 * self.patch * self.patch
        self.Kc = _rup(kc, 64)
        wc = torch.zeros((D, self.Kc), dtype=f32, device=dev)
        wc[:, :kc] = P["encoder.patch_embed.proj.weight"].detach().reshape(D, kc)
        w["conv"], w["conv_b"] = hw(wc, "conv"), c32(P["encoder.patch_embed.proj.bias"])
        w["cls"] = c32(P["encoder.cls_token"].reshape(D))
        w["pos"] = c32(P["encoder.pos_embed"].reshape(-1, D))
        w["norm.w"], w["norm.b"] = c32(P["encoder.norm.weight"]), c32(P["encoder.norm.bias"])
        for i in range(self.layers):
            p, q = f"encoder.blocks.{i}.", f"enc.{i}."
            w[q + "qkv_w"], w[q + "qkv_b"] = hw(P[p + "attn.qkv.weight"], "qkv"), c32(P[p + "attn.qkv.bias"])
            w[q + "out_w"], w[q + "out_b"] = hw(P[p + "attn.proj.weight"], "out"), c32(P[p + "attn.proj.bias"])
            w[q + "fc_w"], w[q + "fc_b"] = hw(P[p + "mlp.fc1.weight"], "fc"), c32(P[p + "mlp.fc1.bias"])
            w[q + "proj_w"], w[q + "proj_b"] = hw(P[p + "mlp.fc2.weight"], "proj"), c32(P[p + "mlp.fc2.bias"])
            for ln, ln2 in (("norm1", "ln1"), ("norm2", "ln2")):
                w[q + ln2 + ".w"], w[q + ln2 + ".b"] = c32(P[p + ln + ".weight"]), c32(P[p + ln + ".bias"])
        self._pack_decoder(w, P, D, self.dec_layers)
        for j in range(3):
            w[f"ffn.{j}.w"], w[f"ffn.{j}.b"] = hw(P[f"ffn.layers.{j}.weight"], "ffn2"), c32(P[f"ffn.layers.{j}.bias"])
        self._w, self._packed_key = w, key
        self._geo.clear()

    def _pos(self, h: int, w: int) -> torch.Tensor:
        """vision_transformer.py:377-401: bicubic `size=` resample of the 28x28 grid (scale = g/h); returned
        unchanged when h*w equals the stored patch COUNT (the reference compares counts only, :385-388)."""
        g = self._geo.get((h, w))
        if g is None:
            if h * w == self.n_pos:
                pos = self._w["pos"]
            else:
                pos = torch.empty((1 + h * w, self.D), dtype=f32, device=self._device())
                ops.posembed_bicubic(self._w["pos"], pos, self.grid, h, w, self.D, np.float32(self.grid) / np.float32(h),
                                     np.float32(self.grid) / np.float32(w), True)
            g = {"pos": pos}
            self._geo_put((h, w), g)
        return g["pos"]

    def forward(self, x: torch.Tensor, inference: bool = False):
        """Returns {"objectness" [B,1,Q,1] (sigmoid), "mask_pred" [B,1,Q,2h,2w]} or, with inference=True,
        {"dts": uint8 [B,H,W] on device, "index": int64 [B]} (selfmask.py:204-224)."""
        self._pack()
        if not (x.is_cuda and x.dtype == f32 and x.dim() == 4 and x.shape[1] == 3):
            raise ZutisHipError("SelfMaskEngine.forward: expected float32 CUDA tensor [B,3,H,W]")
        x = x.contiguous()
        W_, D, p, Q, L = self._w, self.D, self.patch, self.Q, self.dec_layers
        B, _, H, Wd = x.shape
        h, w = (H + p - 1) // p, (Wd + p - 1) // p                                       # make_input_divisible :260-267
        T, R, M = 1 + h * w, B * (1 + h * w), 4 * h * w
        col = self._abuf("col", (B * h * w, self.Kc), self._x3("conv"))
        ops.im2col(x, col, p, self.Kc, pad_to_patch=True)
        pe32 = self._buf("patch_emb", (B * h * w, D), f32)
        self._gemm("conv", col, W_["conv"], pe32, bias=W_["conv_b"])                     # PatchEmbed :182 (conv WITH bias)
        X = self._buf("X", (R, D), f32)
        ops.assemble_tokens_ln(pe32, W_["cls"], self._pos(h, w), None, None, 0.0, X, B, T, D)   # prepare_tokens :269-281
        self._vit_blocks(X, B, T, D, self.heads, self.layers, 1e-6, ops.ACT_GELU_ERF)    # Block :160-170
        tok = self._buf("tok", (B, h * w, D), f32)
        tok16 = self._abuf("tok16", (B * h * w, D), self._x3("dec_kv"))
        ops.layernorm(X, W_["norm.w"], W_["norm.b"], 1e-6, B * h * w, D, out_f32=tok, out_f16=tok16,
                      in_group_rows=h * w, in_group_stride=T, in_offset=1)               # norm(x)[:, 1:]  :298, selfmask.py:94-100
        KALL, VALL = self._decoder_kv(tok16, tok16, B, h * w, D, L)                          # selfmask.py:110-116 (pos=None)
        q16 = self._decoder(KALL, VALL, B, h * w, D, Q, L, self.heads, stack_all=False)
        FEAT = self._abuf("FEAT16", (B * M, D), self._x3("mask"))
        ops.upsample2x_cl(tok, B, h, w, D, out_f16=FEAT)                                 # forward_pixel_decoder :131-135
        masks = torch.empty((B, 1, Q, 2 * h, 2 * w), dtype=f32, device=x.device)
        self._gemm("mask", q16, FEAT, masks, act=ops.ACT_SIGMOID, M=Q, N=M, K=D, lda=D, ldw=D, ldc=M, batch=B,
                   strideA=Q * D, strideW=M * D, strideC=Q * M)                           # einsum("bqn,bnhw->bqhw") + sigmoid :181
        o1 = self._abuf("obj_h1", (B * Q, D), self._x3("ffn2"))
        o2 = self._abuf("obj_h2", (B * Q, D), self._x3("ffn2"))
        self._gemm("ffn2", q16, W_["ffn.0.w"], o1, bias=W_["ffn.0.b"], act=ops.ACT_RELU)  # objectness MLP :182
        self._gemm("ffn2", o1, W_["ffn.1.w"], o2, bias=W_["ffn.1.b"], act=ops.ACT_RELU)
        obj = torch.empty((B, 1, Q, 1), dtype=f32, device=x.device)
        if not inference:
            self._gemm("ffn2", o2, W_["ffn.2.w"], obj, bias=W_["ffn.2.b"], act=ops.ACT_SIGMOID, M=B * Q, N=1, K=D, ldc=1)
            return {"objectness": obj, "mask_pred": masks}
        self._gemm("ffn2", o2, W_["ffn.2.w"], obj, bias=W_["ffn.2.b"], M=B * Q, N=1, K=D, ldc=1)
        # the query with the largest objectness logit is picked on the device (first maximum, as torch.argmax): x4 bilinear of
        # that plane only, cropped to [:H,:W], > 0.5 — no host round trip, so images on different streams overlap
        idx = torch.empty((B,), dtype=torch.int64, device=x.device)
        dts = torch.empty((B, H, Wd), dtype=torch.uint8, device=x.device)
        ops.select_upsample_mask(obj, masks, dts, idx, B, Q, 2 * h, 2 * w, H, Wd, 0.25, 0.25, 0.5)
        return {"dts": dts, "index": idx}
