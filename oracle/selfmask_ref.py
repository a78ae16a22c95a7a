"""ORACLE — test infrastructure, NOT the product path (see oracle/zutis_ref.py header).

CPU restatement of the SelfMask pseudo-labeller (networks/selfmask/selfmask.py:137-245) over the DINO ViT-S/8
encoder (networks/selfmask/vision_transformer.py:97-170,260-304,377-401) and the DETR-style decoder
(networks/selfmask/transformer_decoder.py:104-150,229-297).  Pinned against the real reference through
tests/golden/selfmask_*.npz (oracle/gen_golden.py).
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from . import resample as R
from .zutis_ref import decoder_forward, layer_norm

Tensor = torch.Tensor


def vit_pos_embed(pos_embed: Tensor, h: int, w: int) -> Tensor:
    """vision_transformer.py:377-401: bicubic `size=(h,w)` resample of the 28x28 grid; returned unchanged when
    h*w equals the stored patch count (the reference compares COUNTS only, :385-388)."""
    pe = pos_embed[0]
    n = pe.shape[0] - 1
    if h * w == n:
        return pe
    g = int(math.sqrt(n))
    out = R.bicubic_cl(pe[1:].detach().numpy().reshape(g, g, -1), h, w)
    return torch.cat([pe[:1], torch.from_numpy(out.reshape(h * w, -1))], dim=0)


def dino_vit_forward(P: Dict[str, Tensor], x: Tensor, patch: int = 8, heads: int = 6, prefix: str = "encoder."):
    """vision_transformer.py:269-304 -> (last-layer normed patch tokens [B,hw,D], h, w)."""
    B, _, H, W = x.shape
    pad_w, pad_h = (patch - W % patch) % patch, (patch - H % patch) % patch
    x = F.pad(x, (0, pad_w, 0, pad_h), value=0)                                   # :260-267
    t = F.conv2d(x, P[prefix + "patch_embed.proj.weight"], P[prefix + "patch_embed.proj.bias"], stride=patch)
    h, w = t.shape[-2:]
    t = t.flatten(2).transpose(1, 2)
    D = t.shape[-1]
    t = torch.cat([P[prefix + "cls_token"].expand(B, -1, -1), t], dim=1)
    t = t + vit_pos_embed(P[prefix + "pos_embed"], h, w)[None]
    depth = 1 + max(int(k.split(".")[2]) for k in P if k.startswith(prefix + "blocks."))
    dh = D // heads
    for i in range(depth):                                                         # Block :160-170
        p = f"{prefix}blocks.{i}."
        y = layer_norm(t, P[p + "norm1.weight"], P[p + "norm1.bias"], 1e-6)
        qkv = F.linear(y, P[p + "attn.qkv.weight"], P[p + "attn.qkv.bias"]).reshape(B, -1, 3, heads, dh).permute(2, 0, 3, 1, 4)
        a = torch.softmax(torch.matmul(qkv[0], qkv[1].transpose(-2, -1)) * dh ** -0.5, dim=-1)   # :122-123
        y = torch.matmul(a, qkv[2]).transpose(1, 2).reshape(B, -1, D)
        t = t + F.linear(y, P[p + "attn.proj.weight"], P[p + "attn.proj.bias"])
        y = layer_norm(t, P[p + "norm2.weight"], P[p + "norm2.bias"], 1e-6)
        y = F.gelu(F.linear(y, P[p + "mlp.fc1.weight"], P[p + "mlp.fc1.bias"]))   # exact erf GELU
        t = t + F.linear(y, P[p + "mlp.fc2.weight"], P[p + "mlp.fc2.bias"])
    t = layer_norm(t, P[prefix + "norm.weight"], P[prefix + "norm.bias"], 1e-6)    # :298
    return t[:, 1:], h, w


def selfmask_forward(P: Dict[str, Tensor], x: Tensor, patch: int = 8, heads: int = 6):
    """selfmask.py:137-187 (return_intermediate=False branch): {"objectness" (sigmoid) [B,1,Q,1],
    "mask_pred" [B,1,Q,2h,2w], "objectness_logits" [B,Q]}."""
    B = x.shape[0]
    tok, h, w = dino_vit_forward(P, x, patch, heads)
    D = tok.shape[-1]
    q = decoder_forward(P, tok, None, P["query_embed"], heads, return_intermediate=False)       # [B,Q,D]
    feat = torch.from_numpy(R.bilinear_up2_cl(tok.numpy().reshape(B, h, w, D))).reshape(B, 4 * h * w, D)
    mask = torch.sigmoid(torch.einsum("bqn,bmn->bqm", q, feat)).reshape(B, 1, -1, 2 * h, 2 * w)
    o = F.relu(F.linear(q, P["ffn.layers.0.weight"], P["ffn.layers.0.bias"]))
    o = F.relu(F.linear(o, P["ffn.layers.1.weight"], P["ffn.layers.1.bias"]))
    o = F.linear(o, P["ffn.layers.2.weight"], P["ffn.layers.2.bias"])               # [B,Q,1]
    return {"objectness": torch.sigmoid(o)[:, None], "mask_pred": mask, "objectness_logits": o[..., 0]}


def selfmask_inference(P: Dict[str, Tensor], x: Tensor, patch: int = 8, heads: int = 6, out=None):
    """selfmask.py:204-224: x4 bilinear, crop to the input size, pick the argmax-objectness query, > 0.5 -> uint8.
    (`out`: a selfmask_forward result for the same x, to save the second forward.)"""
    B, _, H, W = x.shape
    if out is None:
        out = selfmask_forward(P, x, patch, heads)
    mp = out["mask_pred"][:, 0].numpy()
    up = R.bilinear_nchw(mp, 4 * mp.shape[2], 4 * mp.shape[3])[..., :H, :W]
    idx = out["objectness_logits"].argmax(dim=1).numpy()
    return [(up[b, idx[b]] > 0.5).astype(np.uint8) for b in range(B)], idx, up
