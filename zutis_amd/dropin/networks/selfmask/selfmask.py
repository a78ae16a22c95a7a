"""Drop-in replacement for the reference's networks/selfmask/selfmask.py (SelfMask) on MI355X.

Same constructor, 267 state_dict keys and forward(x, encoder_only, inference, bilateral_solver) contract
(networks/selfmask/selfmask.py:14-23,137-245); modules are parameter containers only, the arithmetic runs in
zutis_amd.engine.SelfMaskEngine (HIP kernels).  The reference materialises a [B,6,T,T] attention matrix per block
(727 MB at 512x683) and falls back to the CPU on OOM (datasets/index_dataset.py:201-204); flash attention removes both.
"""
from typing import Dict, List

import numpy as np
import torch
import torch.nn as nn

from zutis_amd.engine import SelfMaskEngine


class MLP(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))


class _Block(nn.Module):
    """Keys of vision_transformer.Block (:136-170): norm1, attn.qkv, attn.proj, norm2, mlp.fc1, mlp.fc2."""

    def __init__(self, dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = nn.Module()
        self.attn.qkv = nn.Linear(dim, 3 * dim, bias=True)
        self.attn.proj = nn.Linear(dim, dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = nn.Module()
        self.mlp.fc1 = nn.Linear(dim, 4 * dim)
        self.mlp.fc2 = nn.Linear(4 * dim, dim)


class _ViT(nn.Module):
    """Keys of vision_transformer.VisionTransformer as built by deit_small (:191-258,513-525)."""

    def __init__(self, patch_size=8, embed_dim=384, depth=12, num_heads=6):
        super().__init__()
        self.patch_embed = nn.Module()
        self.patch_embed.proj = nn.Conv2d(3, embed_dim, kernel_size=patch_size, stride=patch_size)
        n = (224 // patch_size) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n + 1, embed_dim))
        nn.init.trunc_normal_(self.pos_embed, std=.02)
        nn.init.trunc_normal_(self.cls_token, std=.02)
        self.blocks = nn.ModuleList(_Block(embed_dim) for _ in range(depth))
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.depth, self.n_embs, self.embed_dim, self.n_heads, self.mlp_ratio, self.patch_size = \
            depth, embed_dim, embed_dim, num_heads, 4, patch_size


class _DecoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead)
        self.multihead_attn = nn.MultiheadAttention(d_model, nhead)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(d_model), nn.LayerNorm(d_model), nn.LayerNorm(d_model)


class _Decoder(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward, num_layers):
        super().__init__()
        self.layers = nn.ModuleList(_DecoderLayer(d_model, nhead, dim_feedforward) for _ in range(num_layers))
        self.norm = nn.LayerNorm(d_model)


class SelfMask(nn.Module):
    def __init__(
            self,
            n_queries: int = 20,
            patch_size: int = 8,
            n_decoder_layers: int = 6,
            normalize_before: bool = False,
            return_intermediate: bool = False,
            scale_factor: int = 2,
            use_binary_classifier: bool = True
    ):
        super(SelfMask, self).__init__()
        if normalize_before or return_intermediate or scale_factor != 2:
            raise NotImplementedError("only the released SelfMask configuration (post-norm, last layer, x2) is on the hot path")
        self.encoder = _ViT(patch_size=patch_size)
        n_dims, n_heads = self.encoder.n_embs, self.encoder.n_heads
        self.decoder = _Decoder(n_dims, n_heads, n_dims * self.encoder.mlp_ratio, n_decoder_layers)
        self.query_embed = nn.Embedding(n_queries, n_dims).weight
        self.ffn = MLP(n_dims, n_dims, 1, num_layers=3)
        self.arch = "vit_small"
        self.use_binary_classifier = use_binary_classifier
        self.scale_factor = scale_factor
        self._engine = None
        self.precision: str = "exact"             # "fast" | "exact" | "f16" (zutis_amd.engine)
        self.cross_attention_key_split: int = 8      # engine_base._decoder / engine_selfmask: 20 queries x ~5500 keys at batch 1 - 8

    def _get_engine(self) -> SelfMaskEngine:
        if self._engine is not None and self._engine.precision != self.precision:
            self._engine = None
        if self._engine is None:
            self._engine = SelfMaskEngine(dict(self.named_parameters()), self.encoder.patch_size, self.encoder.n_heads,
                                          precision=self.precision)
            self._engine.cross_ksplit = self.cross_attention_key_split      # batch-1 evaluation loops: see engine_base._decoder
        return self._engine

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._engine = None
        return out

    def forward(self, x, encoder_only=False, inference: bool = False, bilateral_solver: bool = False):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("SelfMask on MI355X is inference-only: wrap the call in torch.no_grad()")
        if encoder_only:
            raise NotImplementedError("encoder_only=True raises in the reference as well (selfmask.py:162 view error)")
        eng = self._get_engine()
        out = eng.forward(x.float(), inference=inference)
        if not inference:
            return out
        dts_dev = out["dts"]
        dict_outputs: Dict[str, List[torch.Tensor]] = {"dts": [d for d in dts_dev.cpu()]}   # selfmask.py:221-222
        if bilateral_solver:                                                                # selfmask.py:226-237
            # One batched device solve for all images (zh_bilateral_solve_batch) instead of the reference's per-image
            # D2H -> PIL -> NumPy/SciPy round trip; only output 0 of bilateral_solver_output is used there (:230-231), so the
            # hole-filling / labelling post-processing it also runs (and discards) is not reproduced.
            from zutis_amd import ops as _ops
            xf = x.float()
            rgb = torch.stack([_ops.denormalize_u8(xf[b].contiguous()) for b in range(x.shape[0])])   # utils/utils.py:261-273
            soft, _ = _ops.bilateral_solve(rgb, dts_dev.contiguous())
            dict_outputs["dts_bi"] = [d for d in _ops.threshold_f64_u8(soft, 0.5).cpu()]
        return dict_outputs
