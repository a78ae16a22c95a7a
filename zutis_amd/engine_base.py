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

import collections

import math
import os
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



def long_sequence_key_split(items: int, ktiles: int, head_dim: int, x3: bool, out_elems: int) -> int:
    """Key split (1 .. 8) of a LONG self-attention (SelfMask's DINO ViT-S/8 at 512x683: T = 5505, networks/selfmask/vision_transformer.py:110-133)
    from a round-quantisation model of zh_attention_f16's grid: `items` = (image, head, 128-query block) workgroups, each walking `ktiles`
    key tiles.  A CU holds 3 such workgroups (2 for the pipelined split-pair loop and at dh = 96), so 4 images x 6 heads x 44 blocks = 1056
    workgroups are 1.375 rounds of the chip's 768 slots — two rounds, the second a third full — and ONE image (264 workgroups) leaves
    every SIMD a single wave with nothing to overlap its softmax with.  Splitting the keys over S workgroups per item (partials merged by
    attn_combine_kernel) buys finer rounds for one pass over the fp32 partials.  Cost in key-tile times of a full CU:
    rounds x (chunk + fixed) + a last partial round at the (faster) per-tile time of its occupancy + the merge traffic."""
    def wpc_of(n):
        if x3 and (head_dim == 96 or -(-n // 512) <= -(-n // 768)):
            return 2                                       # the launcher's rule for the software-pipelined loop (attention.hip)
        return 3 if head_dim == 64 else 2
    tile_time = {1: 0.78, 2: 0.89, 3: 1.0}                 # per-tile time of a workgroup with 1 / 2 / 3 resident per CU (stamps, profiles/NOTES.md)
    tile_us = 1.85 if x3 else 0.95                         # one key tile of a workgroup at full occupancy
    best, best_cost = 1, None
    for S in range(1, 9):
        chunk = -(-ktiles // S)
        if S > 1 and (S - 1) * chunk >= ktiles:
            continue
        n = items * S
        wpc = wpc_of(n)
        full, rem = divmod(n, 256 * wpc)
        cost = full * (chunk + 2) * tile_time[wpc]
        if rem:
            cost += (chunk + 2) * tile_time[min(wpc, -(-rem // 256))]
        if S > 1:
            cost += S * out_elems * 4 * 2 / 3.0e12 * 1e6 / tile_us      # partials written and read once, ~3 TB/s
        if best_cost is None or cost < best_cost * 0.97:               # a larger split must win by 3 %
            best, best_cost = S, cost
    return best

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
        # captured hipGraphs per input shape: a small LRU of their own (each pins a whole activation-buffer set; they used to share the
        # FIFO of the geometry tables, where a native-resolution set's many shapes evicted the tables and each other)
        self._graphs: "collections.OrderedDict" = collections.OrderedDict()
        self._bufs: Dict[Tuple, torch.Tensor] = {}
        self._pinned: Dict[int, torch.Tensor] = {}    # pinned staging buffers of the predict's one device -> host copy (per engine instance)
        self._buf_gen = 0             # bumped on every (re)allocation: launch plans check it
        self._buf_const: Dict[str, torch.Tensor] = {}   # buffers with constant regions: name -> the tensor that was initialised
        self._pt16_of = self._text16_of = None   # which tensors the cached f16 copies "pt16" / "text16" were made from
        self._status = None           # device status word (ops.STATUS_*): see status_word() / check_finite()

    def fork(self):
        """A second engine over the SAME parameters and packed weights with its own activation buffers: one per HIP stream
        when independent inputs are processed concurrently (buffers are the only mutable state of an engine)."""
        import copy
        self._pack()
        e = copy.copy(self)
        e._bufs, e._buf_gen, e._buf_const = {}, 0, {}
        # input-independent tables are shared; captured graphs are not (they replay into the parent's buffers and stream)
        e._geo = dict(self._geo)
        e._graphs = collections.OrderedDict()
        e._pinned = {}
        e._pt16_of = e._text16_of = None       # provenance of the f16 copies held in the (new, empty) buffer cache
        e._status = None                       # a status word of its own (it is written on the fork's stream)
        return e

    # ------------------------------------------------------------------ the x3 envelope
    # A split pair holds |x| < 65504 with 22 significant bits (min(2^-22 |x|, 3e-8) absolute: below 0.125 the lo half is a subnormal
    # fp16 number).  Inside the envelope the engine is fp32-class; OUTSIDE it must not answer silently: an activation beyond the fp16
    # range stores hi = inf, every product with it is a NaN, and every path of these networks leads into a LayerNorm, whose kernels OR
    # STATUS_NONFINITE into this word when a row's variance is inf / NaN (one compare per row).  The word is sticky; the host reads it
    # where it synchronises anyway (the drop-in's predict) or on request (check_finite) and raises.
    def status_word(self) -> torch.Tensor:
        if self._status is None:
            self._status = torch.zeros((1,), dtype=torch.int32, device=self._device())
        return self._status

    def check_finite(self):
        """Synchronises.  Raises when a forward since the last check met a non-finite activation (fp16 split-pair range exceeded, or an
        inf / NaN in the input or the weights); clears the word."""
        if self._status is None:
            return
        self.raise_on_status(int(self._status.item()))

    def raise_on_status(self, word: int):
        if word & ops.STATUS_NONFINITE:
            if self._status is not None:
                self._status.zero_()
            raise ZutisHipError("non-finite activations: a value left the fp16 split-pair range (|x| >= 65504) or the input / weights hold "
                                "an inf / NaN — the f16x3 engine does not answer outside its envelope (DESIGN.md, Precision)")

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

    _GRAPH_CAP = 6    # captured input shapes kept (least recently replayed goes first); an evicted shape is captured again on its next run

    def _graph_put(self, key, value):
        while len(self._graphs) >= self._GRAPH_CAP:
            self._graphs.popitem(last=False)
        self._graphs[key] = value

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

    def _abuf(self, name: str, shape, split: bool, unit_norm: bool = False) -> Act:
        """Cached fp16 activation buffer, a split pair when a consumer runs in the x3 mode.  unit_norm: rows of norm 1 (elements ~0.04):
        a split pair of them is stored times 2^10 so that its lo half is a normal fp16 number (ops.UNIT_NORM_SCALE; the consumer GEMM
        multiplies the factor out through Act.out_scale)."""
        return Act(self._buf(name, ((2 if split else 1),) + tuple(shape), f16), out_scale=(1.0 / ops.UNIT_NORM_SCALE) if (unit_norm and split) else 1.0)

    @staticmethod
    def _unit_scale_ok(t: torch.Tensor) -> bool:
        """True when `t` times ops.UNIT_NORM_SCALE stays inside the fp16 range with margin (|x| < 32; unit-norm rows are <= 1).
        Synchronises (one max-abs read): callers use it on rare / cached paths only."""
        return float(t.detach().abs().max()) * ops.UNIT_NORM_SCALE < 32768.0

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
        # Few-row regime (batch-1 evaluation: configs/*.yaml val batch_size 1, trainer.py:328-345): the two N = D GEMMs of a block
        # (out_proj, c_proj) are 60 tiles of 128 x 128 for 256 CUs, so their K is split over S workgroups per tile — a batched GEMM
        # over K slabs writing fp32 partial planes — and the planes are summed by the LayerNorm that follows (zh_sum_layernorm_f32:
        # bias + residual + ln_2 / the next block's ln_1 in the same pass).  Same launch count, 4x the workgroups, no LN launch of its own.
        st = self.status_word()
        s_out, s_proj = self._splitk(R, D, D), self._splitk(R, D, Fd)
        sk = self._x3("out") and self._x3("proj") and (s_out > 1 or s_proj > 1)
        parts = self._buf("sk_parts", (max(s_out, s_proj), R, D), f32) if sk else None

        def gemm_parts(site, A, Wt, S):
            K = A.hi.shape[-1]
            return self._gemm(site, A, Wt, parts, M=R, N=D, K=K // S, lda=K, ldw=K, ldc=D, batch=S, strideA=K // S, strideW=K // S, strideC=R * D)
        # Self-attention with few (image, head, 128-query block) items — one image: 12 heads x 10 blocks on 256 CUs — splits the keys
        # over workgroups as the decoder's cross-attention does (zh_attention_f16_splitk + merge): a function of B * heads and T only
        eks, ews = 1, None
        if not causal:
            per_image = heads * -(-T // 128)
            eks = max(1, min(8, 256 // per_image)) if B * per_image <= 128 else 1      # the split itself does not depend on B
            ktiles = -(-T // (32 if xa else 64))
            if eks == 1 and T >= 2048:
                eks = long_sequence_key_split(B * per_image, ktiles, D // heads, xa, B * T * D)
            while eks > 1 and (eks - 1) * -(-ktiles // eks) >= ktiles:
                eks -= 1
            if eks > 1:
                ews = self._buf("enc_attn_ws", (ops.attention_splitk_workspace_size(B, heads, T, D // heads, eks),), torch.uint8)
        for i in range(n_layers):
            pp = f"enc.{i}."
            if not (sk and s_proj > 1 and i > 0):          # split-K regime: ln_1 of block i > 0 came out of block i - 1's last kernel
                ops.layernorm(X, W_[pp + "ln1.w"], W_[pp + "ln1.b"], eps, R, D, out_f16=Y, status=st)
            self._gemm("qkv", Y, W_[pp + "qkv_w"], QKV, bias=W_[pp + "qkv_b"])
            ops.attention(q_, k_, v_, O, batch=B, heads=heads, Tq=T, Tk=T, head_dim=D // heads,
                          ldq=3 * D, ldk=3 * D, ldv=3 * D, ldo=D, strideQ=T * 3 * D, strideK=T * 3 * D, strideV=T * 3 * D,
                          strideO=T * D, causal=causal, x3=xa, ksplit=eks, workspace=ews)
            if sk and s_out > 1:
                gemm_parts("out", O, W_[pp + "out_w"], s_out)
                ops.sum_layernorm(parts, s_out, R, D, bias=W_[pp + "out_b"], residual=X, out_sum=X, gamma=W_[pp + "ln2.w"], beta=W_[pp + "ln2.b"],
                                  eps=eps, out_f16=Y, status=st)
            else:
                self._gemm("out", O, W_[pp + "out_w"], X, bias=W_[pp + "out_b"], residual=X)
                ops.layernorm(X, W_[pp + "ln2.w"], W_[pp + "ln2.b"], eps, R, D, out_f16=Y, status=st)
            self._gemm("fc", Y, W_[pp + "fc_w"], Hh, bias=W_[pp + "fc_b"], act=act)
            if sk and s_proj > 1:
                gemm_parts("proj", Hh, W_[pp + "proj_w"], s_proj)
                nx = f"enc.{i + 1}." if i + 1 < n_layers else None
                ops.sum_layernorm(parts, s_proj, R, D, bias=W_[pp + "proj_b"], residual=X, out_sum=X,
                                  gamma=W_[nx + "ln1.w"] if nx else None, beta=W_[nx + "ln1.b"] if nx else None, eps=eps, out_f16=Y if nx else None,
                                  status=st)
            else:
                self._gemm("proj", Hh, W_[pp + "proj_w"], X, bias=W_[pp + "proj_b"], residual=X)

    # rows (B * T) up to which the N = D GEMMs of a transformer block run split-K; and the split as a function of the shape only
    # (never of the data): results are bitwise reproducible for a given (rows, D, K)
    SPLITK_MAX_ROWS = int(os.environ.get("ZH_SPLITK_MAX_ROWS", "2048"))

    def _splitk(self, R: int, N: int, K: int) -> int:
        """K split of an [R, N] = [R, K] x [N, K]^T GEMM whose partial planes a zh_sum_layernorm_f32 launch adds up.  A function of
        K alone inside the few-row regime: image i's result does not depend on how many images share its batch there."""
        if R > self.SPLITK_MAX_ROWS or K % 64:
            return 1
        s = 1
        while 2 * s <= self.SPLITK_MAX and (K // (2 * s)) % 64 == 0 and K // (2 * s) >= self.SPLITK_MIN_K:
            s *= 2
        return s

    SPLITK_MAX = int(os.environ.get("ZH_SPLITK_MAX", "4"))
    # shortest K slab: c_proj (K = 3072) splits four ways (38.4 -> 25.0 us for the GEMM, + 5 us in the LayerNorm that adds the planes);
    # out_proj (K = 768) does not — its 228 tiles of 64 x 64 already fill the chip (11.8 us; planes + a longer LayerNorm cost more)
    SPLITK_MIN_K = int(os.environ.get("ZH_SPLITK_MIN_K", "512"))

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
        st = self.status_word()
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
        # cross-attention: Q <= 128 queries against M keys is ONE workgroup per (image, head): 8 workgroups at batch 1 (the COCO-20K
        # evaluation's regime), 256 at batch 32 (one per CU, each streaming its K / V with a single tile of prefetch).  The keys can be
        # split over `self.cross_ksplit` workgroups + a merge launch (zh_attention_f16_splitk).  Measured (round 3): batch-1 forward
        # + predict 2.99 / 2.73 / 2.60 / 2.54 ms for splits 1 / 2 / 4 / 8; the batch-32 step with three batches in flight loses
        # 0.3 - 1 % with a split of 2 (2825 / 2842 against 2852 / 2851 images/s, same box: the partials' round trip costs more than
        # the extra occupancy gives there).  The split is a property of the ENGINE INSTANCE (throughput: 1, the engine's default
        # and what bench.py runs; the drop-in modules set 8, they serve batch-1 evaluation loops — at the COCO-20K shape, 480x640 = 4800
        # keys, the forward is 4.25 / 3.56 / 3.04 ms for splits 1 / 2 / 8) and never of the batch — image
        # i's result is bitwise independent of its position in the batch and of the other images, and bitwise equal across batch sizes that
        # fall on the same side of the shape thresholds of DESIGN 3b (split-K rows, key-split items, skinny rows); across a threshold the
        # sums are re-associated: ~1e-7 on tokens, ~5e-7 on masks (tests/test_e2e_gpu.py::test_batch_invariance_full_size).
        cs = self.cross_ksplit
        if cs == "auto":
            # opt-in (round 6; bench.py's config-4 runs: 8 images x 8 heads = 64 workgroups of 171 key tiles on 256 CUs): the split that
            # puts about one workgroup on every CU.  It depends on the BATCH, so results are re-associated between batch sizes (~1e-7) —
            # which is why it is not the default: equal rank shards must reproduce the single-GPU batch bit for bit
            cs = max(1, min(8, 256 // max(1, B * heads)))
        ksplit = cs if (Q <= 128 and M >= 1024) else 1
        ktiles = -(-M // (32 if xk else 64))               # key tiles of the kernel (32 keys for split pairs, 64 for fp16)
        while ksplit > 1 and (ksplit - 1) * -(-ktiles // ksplit) >= ktiles:
            ksplit -= 1                                    # the largest split that leaves no workgroup without keys (a function of M only)
        attn_ws = None
        if ksplit > 1:
            attn_ws = self._buf("attn_ws", (ops.attention_splitk_workspace_size(B, heads, Q, dh, ksplit),), torch.uint8)
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
            ops.layernorm(t1, W_[pp + "norm1.w"], W_[pp + "norm1.b"], 1e-5, R, D, out_f32=norm_out32, out_f16=norm_out16, status=st)
        # tgt = zeros (zutis.py:164) and query_pos is a parameter, so layer 0's whole self-attention block — projections of
        # (0 + query_pos, 0), attention over the Q queries, out-projection, norm1 — does not depend on the image: its result
        # (tgt after norm1, fp32 and fp16; [R, D] = the same Q rows for every image) and the cross-attention's query projection of
        # it (:281-282) are computed once per (batch rows, parameter version) with the same kernels and cached; layer 0 then
        # starts at the cross-attention itself.
        ikey = ("dec_init", R, self._packed_key)
        init = self._geo.get(ikey)
        if init is None:
            dev = self._device()
            z16 = Act(torch.zeros((2 if xd else 1, R, D), dtype=f16, device=dev))
            init = {"tgt0": torch.empty((R, D), dtype=f32, device=dev), "tgt0_16": Act.empty((R, D), xd, dev), "qc0_16": Act.empty((R, D), xd, dev)}
            self_attention_block("dec.0.", z16, None, init["tgt0"], init["tgt0_16"])      # q = k = query_pos, v = 0, + tgt (= 0)
            self._gemm("dec", init["tgt0_16"], W_["dec.0.ca_q_w"], init["qc0_16"], residual=W_["dec.0.ca_q_tab"], res_rows=Q)
            self._geo_put(ikey, init)
        for l in range(L):
            pp = f"dec.{l}."
            if l == 0:
                tgt_in, qc = init["tgt0"], init["qc0_16"]
            else:
                self_attention_block(pp, tgt16, tgt, tgt, tgt16)                            # transformer.py:272-278
                tgt_in, qc = tgt, qc16
                self._gemm("dec", tgt16, W_[pp + "ca_q_w"], qc16, residual=W_[pp + "ca_q_tab"], res_rows=Q)   # :281-282 query projection
            ops.attention(qc, KALL.view(KALL.hi[:, l * D:]), VALL.view(VALL.hi[:, l * D:]), o16, batch=B, heads=heads, Tq=Q, Tk=M,
                          head_dim=dh, ldq=D, ldk=L * D, ldv=L * D, ldo=D, strideQ=Q * D, strideK=M * L * D, strideV=M * L * D,
                          strideO=Q * D, x3=xk, ksplit=ksplit, workspace=attn_ws)
            self._gemm("dec", o16, W_[pp + "ca_o_w"], t1, bias=W_[pp + "ca_o_b"], residual=tgt_in)
            ops.layernorm(t1, W_[pp + "norm2.w"], W_[pp + "norm2.b"], 1e-5, R, D, out_f32=tgt, out_f16=tgt16, status=st)
            self._gemm("dec", tgt16, W_[pp + "l1_w"], ff16, bias=W_[pp + "l1_b"], act=ops.ACT_RELU)
            # linear2 (:289-290; K = 2048): in the few-row regime its K is split over workgroups (a batched GEMM over K slabs) and the
            # planes meet in the LayerNorm below, with the bias and the residual (20.9 -> ~10 us for 100 queries)
            s_l2 = self._splitk(R, D, Ff) if xd else 1
            if s_l2 > 1:
                l2p = self._buf("dec_l2_parts", (s_l2, R, D), f32)
                self._gemm("dec", ff16, W_[pp + "l2_w"], l2p, M=R, N=D, K=Ff // s_l2, lda=Ff, ldw=Ff, ldc=D, batch=s_l2, strideA=Ff // s_l2,
                           strideW=Ff // s_l2, strideC=R * D)
                src = dict(parts=l2p, n_parts=s_l2, bias=W_[pp + "l2_b"], residual=tgt)
            else:
                self._gemm("dec", ff16, W_[pp + "l2_w"], t1, bias=W_[pp + "l2_b"], residual=tgt)
                src = dict(parts=t1, n_parts=1)
            # norm3 (:291) and, where the layer's output is kept, decoder.norm on top of it (:140-150; stacked [B,L,Q,D]) in ONE pass
            n3 = dict(gamma=W_[pp + "norm3.w"], beta=W_[pp + "norm3.b"], eps=1e-5, out_f32=tgt, out_f16=tgt16)
            if stack_all:
                n3.update(gamma2=W_["dec.norm.w"], beta2=W_["dec.norm.b"], eps2=1e-5, out2_f16=inter16, out2_group_rows=Q, out2_group_stride=L * Q,
                          out2_offset=l * Q)
            elif l == L - 1:
                n3.update(gamma2=W_["dec.norm.w"], beta2=W_["dec.norm.b"], eps2=1e-5, out2_f16=inter16, out2_f32=out32)
            ops.sum_layernorm(src.pop("parts"), src.pop("n_parts"), R, D, **src, **n3, status=st)
        return inter16

    cross_ksplit = int(os.environ.get("ZH_CROSS_KSPLIT", "1"))      # class default (env = developer override); instances may set it

    _dec_out_sites = ("ffn2",)   # sites consuming the decoder's normed outputs (ZUTIS: ffn2; SelfMask: mask einsum + objectness MLP)
