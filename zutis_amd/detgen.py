"""Deterministic, torch-free synthetic weights and inputs.

Every tensor is a pure function of (seed, name, shape): a splitmix64 hash of a
per-element counter, folded to the sum of four 16-bit uniforms (Irwin-Hall,
variance-corrected to N(0,1)-like).  Only integer ops and one float64 multiply
are used, so the numbers are bit-identical on every box.  Golden fixtures store
only *outputs*; tests, bench.py and oracle/gen_golden.py regenerate inputs and
weights from this module.

Shapes follow the reference state_dict: 275 keys for ViT-B/16.
Init scales follow networks/clip_arch.py:342-350,507-514 (CLIP init) and torch
defaults for the DETR-style decoder (networks/transformer.py:231-251).
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return x ^ (x >> np.uint64(31))


def det_normal(name: str, shape, std: float = 1.0, mean: float = 0.0, seed: int = 0) -> np.ndarray:
    """float32 array ~ N(mean, std^2) (Irwin-Hall n=4), deterministic in (seed, name, index)."""
    n = int(np.prod(shape)) if len(shape) else 1
    base = np.uint64((_fnv1a64(name) ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        ctr = (np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + base) & _MASK
    h = _splitmix64(ctr)
    m16 = np.uint64(0xFFFF)
    s = ((h & m16) + ((h >> np.uint64(16)) & m16) + ((h >> np.uint64(32)) & m16) + (h >> np.uint64(48))).astype(np.int64)
    # sum of 4 U{0..65535}: mean 2*65535, var 4*(65536^2-1)/12
    z = (s - 2 * 65535).astype(np.float64) * (1.0 / np.sqrt(4.0 * (65536.0 ** 2 - 1.0) / 12.0))
    return (z * std + mean).astype(np.float32).reshape(shape)


@dataclass(frozen=True)
class ZutisConfig:
    """Architecture hyper-parameters (reference: networks/zutis.py:16-138, clip_arch.py:590-627)."""
    width: int = 768          # encoder.width
    layers: int = 12          # encoder.transformer.layers
    patch: int = 16           # conv1 kernel = stride
    grid: int = 14            # input_resolution // patch (pos-embed grid)
    embed_dim: int = 512      # encoder.proj output (= text dim)
    n_queries: int = 100
    dec_layers: int = 6
    dec_heads: int = 8
    dec_ff: int = 2048
    ffn_hidden: int = 256

    @property
    def heads(self) -> int:   # clip_arch.py:606 vision_heads = vision_width // 64
        return self.width // 64


VIT_B16 = ZutisConfig()
VIT_B32 = ZutisConfig(patch=32, grid=7)
# tiny config used for the committed end-to-end golden fixtures
# (encoder dh=64 with 3 heads, decoder dh=96 with 2 heads; dec_ff/ffn_hidden are fixed by the reference ctor)
TINY = ZutisConfig(width=192, layers=2, patch=16, grid=4, embed_dim=64, n_queries=5,
                   dec_layers=2, dec_heads=2, dec_ff=2048, ffn_hidden=256)


# config of the build_model / convert_weights fixture (tests/golden/a4_build_model.npz): dh = 64 in encoder and decoder
A4_TINY = ZutisConfig(width=128, layers=2, patch=16, grid=4, embed_dim=64, n_queries=5, dec_layers=2, dec_heads=2)


def zutis_param_shapes(cfg: ZutisConfig) -> "OrderedDict[str, Tuple[Tuple[int, ...], float, float]]":
    """name -> (shape, std, mean) in reference state_dict order-insensitive form."""
    D, L = cfg.width, cfg.layers
    attn_std = D ** -0.5
    proj_std = (D ** -0.5) * ((2 * L) ** -0.5)
    fc_std = (2 * D) ** -0.5
    P: "OrderedDict[str, Tuple[Tuple[int, ...], float, float]]" = OrderedDict()
    P["query_embed"] = ((cfg.n_queries, D), 1.0, 0.0)
    P["encoder.class_embedding"] = ((D,), D ** -0.5, 0.0)
    P["encoder.positional_embedding"] = ((cfg.grid * cfg.grid + 1, D), D ** -0.5, 0.0)
    P["encoder.proj"] = ((D, cfg.embed_dim), D ** -0.5, 0.0)
    P["encoder.conv1.weight"] = ((D, 3, cfg.patch, cfg.patch), (3 * cfg.patch * cfg.patch) ** -0.5, 0.0)
    for ln in ("ln_pre", "ln_post"):
        P[f"encoder.{ln}.weight"] = ((D,), 0.1, 1.0)
        P[f"encoder.{ln}.bias"] = ((D,), 0.1, 0.0)
    for i in range(L):
        p = f"encoder.transformer.resblocks.{i}."
        P[p + "attn.in_proj_weight"] = ((3 * D, D), attn_std, 0.0)
        P[p + "attn.in_proj_bias"] = ((3 * D,), 0.02, 0.0)
        P[p + "attn.out_proj.weight"] = ((D, D), proj_std, 0.0)
        P[p + "attn.out_proj.bias"] = ((D,), 0.02, 0.0)
        P[p + "ln_1.weight"] = ((D,), 0.1, 1.0)
        P[p + "ln_1.bias"] = ((D,), 0.1, 0.0)
        P[p + "mlp.c_fc.weight"] = ((4 * D, D), fc_std, 0.0)
        P[p + "mlp.c_fc.bias"] = ((4 * D,), 0.02, 0.0)
        P[p + "mlp.c_proj.weight"] = ((D, 4 * D), proj_std, 0.0)
        P[p + "mlp.c_proj.bias"] = ((D,), 0.02, 0.0)
        P[p + "ln_2.weight"] = ((D,), 0.1, 1.0)
        P[p + "ln_2.bias"] = ((D,), 0.1, 0.0)
    dims = [D, cfg.ffn_hidden, cfg.ffn_hidden, D]
    for ffn in ("ffn1", "ffn2"):
        for j in range(3):
            P[f"{ffn}.layers.{j}.weight"] = ((dims[j + 1], dims[j]), (2.0 / (dims[j] + dims[j + 1])) ** 0.5, 0.0)
            P[f"{ffn}.layers.{j}.bias"] = ((dims[j + 1],), 0.05, 0.0)
    F = cfg.dec_ff
    for i in range(cfg.dec_layers):
        p = f"decoder.layers.{i}."
        for a in ("self_attn", "multihead_attn"):
            P[p + a + ".in_proj_weight"] = ((3 * D, D), (2.0 / (4 * D)) ** 0.5, 0.0)
            P[p + a + ".in_proj_bias"] = ((3 * D,), 0.02, 0.0)
            P[p + a + ".out_proj.weight"] = ((D, D), D ** -0.5, 0.0)
            P[p + a + ".out_proj.bias"] = ((D,), 0.02, 0.0)
        P[p + "linear1.weight"] = ((F, D), (2.0 / (D + F)) ** 0.5, 0.0)
        P[p + "linear1.bias"] = ((F,), 0.02, 0.0)
        P[p + "linear2.weight"] = ((D, F), (2.0 / (D + F)) ** 0.5, 0.0)
        P[p + "linear2.bias"] = ((D,), 0.02, 0.0)
        for n in ("norm1", "norm2", "norm3"):
            P[p + n + ".weight"] = ((D,), 0.1, 1.0)
            P[p + n + ".bias"] = ((D,), 0.1, 0.0)
    P["decoder.norm.weight"] = ((D,), 0.1, 1.0)
    P["decoder.norm.bias"] = ((D,), 0.1, 0.0)
    return P


def zutis_state_dict(cfg: ZutisConfig, seed: int = 1234) -> "OrderedDict[str, np.ndarray]":
    """Deterministic float32 numpy state_dict with the reference's keys/shapes."""
    return OrderedDict((k, det_normal(k, shp, std, mean, seed))
                       for k, (shp, std, mean) in zutis_param_shapes(cfg).items())


C3_THRESHOLD = 0.7      # mask threshold of the config-3 fixture (predict's `threshold` argument, zutis.py:345)


def c3_state_dict(cfg: ZutisConfig, seed: int = 1234) -> "OrderedDict[str, np.ndarray]":
    """zutis_state_dict() with a decoder that tells its queries apart, for the config-3 (instance + NMS) fixture: at random init
    every query produces the same all-foreground mask of the same class (IoU 0.999 between any two: hard NMS leaves ONE
    survivor).  Larger query embeddings, sharper decoder attention (q / k rows x8) and a wider ffn2 give masks with a mean
    pairwise IoU of ~0.45 at threshold C3_THRESHOLD: ~9 categories, ~100 candidates -> 12-17 hard-NMS survivors per image."""
    sd = zutis_state_dict(cfg, seed)
    D = cfg.width
    sd["query_embed"] = sd["query_embed"] * np.float32(20.0)
    for i in range(cfg.dec_layers):
        for att in ("multihead_attn", "self_attn"):
            w = sd[f"decoder.layers.{i}.{att}.in_proj_weight"].copy()
            w[:2 * D] *= np.float32(8.0)
            sd[f"decoder.layers.{i}.{att}.in_proj_weight"] = w
    for j in range(3):
        sd[f"ffn2.layers.{j}.weight"] = sd[f"ffn2.layers.{j}.weight"] * np.float32(3.0)
    return sd


STRESS_FIXED_CHANNELS = (7, 300, 611)     # same large value at every token ("outlier feature dimensions")
STRESS_TOKEN_CHANNEL = 123                # large, token-dependent value ("massive activations")


def stress_state_dict(cfg: ZutisConfig, magnitude: float = 100.0, seed: int = 1234) -> "OrderedDict[str, np.ndarray]":
    """zutis_state_dict() with the residual-stream outliers real CLIP checkpoints have and random init lacks
    (SURVEY.md §7 "Precision vs roofline"): from encoder block 0 on, three residual channels carry a constant of
    +-`magnitude` (x the ~unit scale of the other channels) at every token, and from block 1 on one more channel carries a
    token-dependent value of that scale.  LayerNorm gains are left near 1 (no attenuation: the worst case), weights are
    generic fp32 (not fp16-representable), and the encoder's last block gets ordinary weights so the outliers also
    reach ln_post and the head."""
    sd = zutis_state_dict(cfg, seed)
    D = cfg.width
    ch = [c % D for c in STRESS_FIXED_CHANNELS]
    b = sd["encoder.transformer.resblocks.0.mlp.c_proj.bias"].copy()
    for j, c in enumerate(ch):
        b[c] += magnitude * (1.0 if j % 2 == 0 else -1.0)
    sd["encoder.transformer.resblocks.0.mlp.c_proj.bias"] = b
    if cfg.layers > 1:
        w = sd["encoder.transformer.resblocks.1.mlp.c_proj.weight"].copy()
        # c_proj output std at init ~ proj_std * sqrt(4D) * rms(quickgelu(h)) ~ 0.25: scale one row to ~magnitude / 2
        w[STRESS_TOKEN_CHANNEL % D] *= np.float32(2.0 * magnitude)
        sd["encoder.transformer.resblocks.1.mlp.c_proj.weight"] = w
    return sd


def selfmask_param_shapes(n_queries: int = 20, D: int = 384, depth: int = 12, patch: int = 8, dec_layers: int = 6,
                          pos_grid: int = 28):
    """SelfMask state_dict (267 keys): DINO ViT-S/8 encoder + 6-layer decoder + objectness MLP
    (networks/selfmask/selfmask.py:14-48, vision_transformer.py:191-258,513-525)."""
    P = OrderedDict()
    P["query_embed"] = ((n_queries, D), 1.0, 0.0)
    P["encoder.cls_token"] = ((1, 1, D), 0.02, 0.0)
    P["encoder.pos_embed"] = ((1, pos_grid * pos_grid + 1, D), 0.02, 0.0)
    P["encoder.patch_embed.proj.weight"] = ((D, 3, patch, patch), (3 * patch * patch) ** -0.5, 0.0)
    P["encoder.patch_embed.proj.bias"] = ((D,), 0.02, 0.0)
    for i in range(depth):
        p = f"encoder.blocks.{i}."
        for n in ("norm1", "norm2"):
            P[p + n + ".weight"] = ((D,), 0.1, 1.0)
            P[p + n + ".bias"] = ((D,), 0.1, 0.0)
        P[p + "attn.qkv.weight"] = ((3 * D, D), D ** -0.5, 0.0)
        P[p + "attn.qkv.bias"] = ((3 * D,), 0.02, 0.0)
        P[p + "attn.proj.weight"] = ((D, D), (D ** -0.5) * ((2 * depth) ** -0.5), 0.0)
        P[p + "attn.proj.bias"] = ((D,), 0.02, 0.0)
        P[p + "mlp.fc1.weight"] = ((4 * D, D), (2 * D) ** -0.5, 0.0)
        P[p + "mlp.fc1.bias"] = ((4 * D,), 0.02, 0.0)
        P[p + "mlp.fc2.weight"] = ((D, 4 * D), (D ** -0.5) * ((2 * depth) ** -0.5), 0.0)
        P[p + "mlp.fc2.bias"] = ((D,), 0.02, 0.0)
    P["encoder.norm.weight"] = ((D,), 0.1, 1.0)
    P["encoder.norm.bias"] = ((D,), 0.1, 0.0)
    F = 4 * D
    for i in range(dec_layers):
        p = f"decoder.layers.{i}."
        for a in ("self_attn", "multihead_attn"):
            P[p + a + ".in_proj_weight"] = ((3 * D, D), (2.0 / (4 * D)) ** 0.5, 0.0)
            P[p + a + ".in_proj_bias"] = ((3 * D,), 0.02, 0.0)
            P[p + a + ".out_proj.weight"] = ((D, D), D ** -0.5, 0.0)
            P[p + a + ".out_proj.bias"] = ((D,), 0.02, 0.0)
        P[p + "linear1.weight"] = ((F, D), (2.0 / (D + F)) ** 0.5, 0.0)
        P[p + "linear1.bias"] = ((F,), 0.02, 0.0)
        P[p + "linear2.weight"] = ((D, F), (2.0 / (D + F)) ** 0.5, 0.0)
        P[p + "linear2.bias"] = ((D,), 0.02, 0.0)
        for n in ("norm1", "norm2", "norm3"):
            P[p + n + ".weight"] = ((D,), 0.1, 1.0)
            P[p + n + ".bias"] = ((D,), 0.1, 0.0)
    P["decoder.norm.weight"] = ((D,), 0.1, 1.0)
    P["decoder.norm.bias"] = ((D,), 0.1, 0.0)
    dims = [D, D, D, 1]
    for j in range(3):
        P[f"ffn.layers.{j}.weight"] = ((dims[j + 1], dims[j]), (2.0 / (dims[j] + dims[j + 1])) ** 0.5, 0.0)
        P[f"ffn.layers.{j}.bias"] = ((dims[j + 1],), 0.05, 0.0)
    return P


def selfmask_state_dict(seed: int = 4321) -> "OrderedDict[str, np.ndarray]":
    return OrderedDict((k, det_normal("selfmask." + k, shp, std, mean, seed))
                       for k, (shp, std, mean) in selfmask_param_shapes().items())


@dataclass(frozen=True)
class ClipTextConfig:
    """CLIP text tower hyper-parameters (networks/clip_arch.py:443-447,484-495; ViT-B/16 + ViT-B/32 checkpoints: ctx 77,
    vocab 49408, width 512, 8 heads = width // 64, 12 layers, embed 512)."""
    context_length: int = 77
    vocab_size: int = 49408
    width: int = 512
    layers: int = 12
    embed_dim: int = 512

    @property
    def heads(self) -> int:
        return self.width // 64


TEXT_B = ClipTextConfig()
TEXT_TINY = ClipTextConfig(context_length=12, vocab_size=96, width=128, layers=2, embed_dim=64)


def clip_text_param_shapes(cfg: ClipTextConfig):
    """CLIP state_dict keys of the text tower (clip_arch.py:484-495); stds follow CLIP.initialize_parameters (:497-523)."""
    D, L = cfg.width, cfg.layers
    attn_std, proj_std, fc_std = D ** -0.5, (D ** -0.5) * ((2 * L) ** -0.5), (2 * D) ** -0.5
    P = OrderedDict()
    P["token_embedding.weight"] = ((cfg.vocab_size, D), 0.02, 0.0)
    P["positional_embedding"] = ((cfg.context_length, D), 0.01, 0.0)
    for i in range(L):
        p = f"transformer.resblocks.{i}."
        P[p + "attn.in_proj_weight"] = ((3 * D, D), attn_std, 0.0)
        P[p + "attn.in_proj_bias"] = ((3 * D,), 0.02, 0.0)
        P[p + "attn.out_proj.weight"] = ((D, D), proj_std, 0.0)
        P[p + "attn.out_proj.bias"] = ((D,), 0.02, 0.0)
        P[p + "ln_1.weight"] = ((D,), 0.1, 1.0)
        P[p + "ln_1.bias"] = ((D,), 0.1, 0.0)
        P[p + "mlp.c_fc.weight"] = ((4 * D, D), fc_std, 0.0)
        P[p + "mlp.c_fc.bias"] = ((4 * D,), 0.02, 0.0)
        P[p + "mlp.c_proj.weight"] = ((D, 4 * D), proj_std, 0.0)
        P[p + "mlp.c_proj.bias"] = ((D,), 0.02, 0.0)
        P[p + "ln_2.weight"] = ((D,), 0.1, 1.0)
        P[p + "ln_2.bias"] = ((D,), 0.1, 0.0)
    P["ln_final.weight"] = ((D,), 0.1, 1.0)
    P["ln_final.bias"] = ((D,), 0.1, 0.0)
    P["text_projection"] = ((D, cfg.embed_dim), D ** -0.5, 0.0)
    return P


def clip_text_state_dict(cfg: ClipTextConfig, seed: int = 2468):
    return OrderedDict((k, det_normal("text." + k, shp, std, mean, seed)) for k, (shp, std, mean) in clip_text_param_shapes(cfg).items())


def clip_full_state_dict(cfg: ZutisConfig, seed: int = 97) -> "OrderedDict[str, np.ndarray]":
    """A complete CLIP state_dict in the third-party package's key layout (visual.* + text tower) carrying GENERIC fp32
    values (not fp16-representable): the input of build_model / convert_weights (clip_arch.py:566-627).  The text tower is
    minimal (ctx 8, vocab 64, width 64, 1 layer) — only its key set matters to build_model's architecture inference."""
    sd = OrderedDict()
    for k, (shp, std, mean) in zutis_param_shapes(cfg).items():
        if k.startswith("encoder."):
            sd["visual." + k[len("encoder."):]] = det_normal("clipfull." + k, shp, std, mean, seed)
    tc = ClipTextConfig(context_length=8, vocab_size=64, width=64, layers=1, embed_dim=cfg.embed_dim)
    for k, (shp, std, mean) in clip_text_param_shapes(tc).items():
        sd[k] = det_normal("clipfull.text." + k, shp, std, mean, seed)
    return sd


def text_tokens(n: int, cfg: ClipTextConfig, seed: int = 11) -> np.ndarray:
    """int64 [n, ctx] shaped like clip.tokenize output: SOT (= vocab-2), 1 .. ctx-2 body tokens, EOT (= vocab-1, the
    unique maximum: encode_text locates it with argmax, clip_arch.py:545), zero padding."""
    u = np.abs(det_normal("tokens", (n, cfg.context_length), 1.0, 0.0, seed))
    body = 1 + (u * 1e6).astype(np.int64) % (cfg.vocab_size - 3)
    lens = 1 + (np.abs(det_normal("token_lens", (n,), 1.0, 0.0, seed)) * 1e6).astype(np.int64) % (cfg.context_length - 2)
    out = np.zeros((n, cfg.context_length), dtype=np.int64)
    for i in range(n):
        L = int(lens[i])
        out[i, 0] = cfg.vocab_size - 2
        out[i, 1:1 + L] = body[i, :L]
        out[i, 1 + L] = cfg.vocab_size - 1
    return out


def text_embeddings(n_categories: int, dim: int, seed: int = 7) -> np.ndarray:
    """Unit-norm rows standing in for CLIP text embeddings (networks/zutis.py:36-37)."""
    t = det_normal("text_embeddings", (n_categories, dim), seed=seed).astype(np.float64)
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    return t.astype(np.float32)


def images(b: int, h: int, w: int, seed: int = 0) -> np.ndarray:
    """Synthetic normalised images ~N(0,1), [b,3,h,w] float32."""
    return det_normal(f"images_{b}x3x{h}x{w}", (b, 3, h, w), seed=seed)


def selfmask_like_rgb(h: int, w: int, seed: int = 3) -> np.ndarray:
    """u8 RGB H×W×3: smooth gradient + blobs + 5 % noise (bilateral-solver input)."""
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    img = np.stack([40 + 150 * xx / max(w - 1, 1), 60 + 120 * yy / max(h - 1, 1),
                    200 - 100 * (xx + yy) / max(h + w - 2, 1)], axis=-1)
    c = det_normal("blob_centres", (6, 5), seed=seed).astype(np.float64)
    for k in range(6):
        cy, cx = (0.5 + 0.25 * c[k, 0]) * h, (0.5 + 0.25 * c[k, 1]) * w
        r = (0.08 + 0.04 * abs(c[k, 2])) * min(h, w)
        m = ((yy - cy) ** 2 + (xx - cx) ** 2) < r * r
        img[m] = np.clip(np.array([128.0, 128.0, 128.0]) + 60 * c[k, 2:5], 0, 255)
    noise = det_normal("rgb_noise", (h, w, 3), seed=seed).astype(np.float64)
    sel = det_normal("rgb_noise_sel", (h, w, 1), seed=seed) > 1.645  # ≈5 %
    img = np.where(sel, img + 40 * noise, img)
    return np.clip(img, 0, 255).astype(np.uint8)
