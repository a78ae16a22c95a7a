"""SelfMaskEngine: the SelfMask pseudo-labeller on the HIP path (networks/selfmask/selfmask.py:137-245, DINO ViT-S/8 encoder
networks/selfmask/vision_transformer.py:260-304).  Shared kernel sequences: zutis_amd/engine_base.py."""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

from . import ops
from ._lib import ZutisHipError
from .engine_base import _EngineBase, _rup, f16, f32


class SelfMaskEngine(_EngineBase):
    """SelfMask pseudo-labeller (networks/selfmask/selfmask.py:137-245): DINO ViT-S/8 encoder
    (vision_transformer.py:260-304) -> 6-layer decoder, 20 queries, no memory pos -> x2 upsampled tokens . queries ->
    objectness MLP; inference picks the argmax-objectness query, x4 bilinear, crop, > 0.5."""

    _dec_out_sites = ("mask", "ffn2")
    # 20 queries against ~5500 memory tokens, 6 heads, batches of 1 - 8: the cross-attention is 6 .. 48 workgroups of 172 key tiles
    # each; its keys are split 8 ways (engine_base._decoder: a property of the engine, never of the batch)
    cross_ksplit = 8

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
        self._graphs.clear()

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
                      in_group_rows=h * w, in_group_stride=T, in_offset=1, status=self.status_word())               # norm(x)[:, 1:]  :298, selfmask.py:94-100
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
