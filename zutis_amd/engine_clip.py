"""CLIP towers on the HIP path: ClipImageEncoder (`encode_image`, utils/extract_image_embeddings.py:72-73 over
networks/clip_arch.py:413-431,531-532) and ClipTextEncoder (`encode_text`, clip_arch.py:534-547, + the prompt ensembling of
utils/extract_text_embeddings.py:98-115).  Shared kernel sequences: zutis_amd/engine_base.py."""
from __future__ import annotations

import math
from typing import Dict

import torch

from . import ops
from ._lib import ZutisHipError
from .engine_base import _EngineBase, f16, f32


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
                      in_group_rows=1, in_group_stride=1 + h * w, in_offset=0, status=self.status_word())              # ln_post(x[:, 0, :])
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
        ops.layernorm(eot, W_["lnf.w"], W_["lnf.b"], 1e-5, n, D, out_f16=e16, status=self.status_word())          # :541 ln_final
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
