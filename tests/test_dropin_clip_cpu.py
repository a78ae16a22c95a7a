"""The drop-in constructor's `clip.load` branch and `update_text_embeddings` (reference networks/zutis.py:35-57, 333-338) with a
stand-in `clip` package (the real one is absent from this image): what is asked of the package, what is done with what it returns.
CPU only — the constructor builds parameter containers; no kernel runs."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zutis_amd", "dropin"))
from zutis_amd import detgen  # noqa: E402


class _StubClipModel:
    """What `clip.load` hands back, as far as the constructor uses it: encode_text() (fp16 rows, NOT normalised) and state_dict()."""

    def __init__(self, cfg, calls):
        self.cfg, self.calls = cfg, calls
        self.sd = {k: torch.from_numpy(v) for k, v in detgen.clip_full_state_dict(cfg).items()}

    def encode_text(self, tokens):
        self.calls.append(("encode_text", tuple(tokens.shape), str(tokens.device)))
        n = tokens.shape[0]
        rows = torch.from_numpy(detgen.det_normal("stubclip.text", (n, self.cfg.embed_dim), 3.0, 0.5, int(tokens.sum()) % 97))
        return rows.to(torch.float16)                    # clip's towers are fp16 on a GPU: the constructor casts back (zutis.py:36)

    def state_dict(self):
        return self.sd


def _install(monkeypatch, cfg):
    import types
    import networks.zutis as NZ
    calls = []
    stub = types.ModuleType("clip")

    def load(name, device=None):
        calls.append(("load", name, str(device)))
        return _StubClipModel(cfg, calls), None

    def tokenize(texts):
        calls.append(("tokenize", tuple(texts)))
        t = torch.zeros((len(texts), 8), dtype=torch.long)
        for i, s in enumerate(texts):
            t[i, 0], t[i, 1], t[i, -1] = 62, 1 + (sum(map(ord, s)) % 60), 63
        return t
    stub.load, stub.tokenize = load, tokenize
    monkeypatch.setattr(NZ, "_clip", stub)
    return NZ, calls


def test_constructor_through_clip_load(monkeypatch):
    cfg = detgen.A4_TINY
    NZ, calls = _install(monkeypatch, cfg)
    cats = ["cat", "dog", "a very long category name"]
    dev = torch.device("cpu")
    net = NZ.ZUTIS(categories=cats, clip_arch="dilatedViT-B/16", n_queries=cfg.n_queries, n_decoder_layers=cfg.dec_layers, n_heads=cfg.dec_heads,
                   device=dev)
    # zutis.py:35-37: load(arch without the "dilated" prefix) -> tokenize(categories) -> encode_text -> float32 -> row-normalise
    assert calls[0] == ("load", "ViT-B/16", "cpu") and calls[1] == ("tokenize", tuple(cats)) and calls[2][0] == "encode_text"
    te = net.text_embeddings
    assert te.dtype == torch.float32 and te.shape == (3, cfg.embed_dim) and not te.requires_grad
    assert torch.allclose(te.norm(dim=1), torch.ones(3), atol=1e-6)
    raw = _StubClipModel(cfg, []).encode_text(NZ._clip.tokenize(cats)).float()
    assert torch.equal(te, raw / raw.norm(dim=1, keepdim=True))
    assert list(net.category_to_text_embedding) == cats and torch.equal(net.category_to_text_embedding["dog"], te[1])
    assert net.n_dims_text == cfg.embed_dim
    # the visual tower: architecture inferred from the package's state_dict keys (clip_arch.py:595-600), values as convert_weights
    # leaves them (fp16-rounded conv / Linear / attention / proj; LayerNorm and embeddings untouched), then .float() (zutis.py:55)
    enc = net.encoder
    assert [enc.width, enc.transformer.layers, enc.patch_size, enc.input_resolution, enc.output_dim] == \
        [cfg.width, cfg.layers, cfg.patch, cfg.patch * cfg.grid, cfg.embed_dim]
    src = {k: torch.from_numpy(v) for k, v in detgen.clip_full_state_dict(cfg).items()}
    sd = enc.state_dict()
    assert torch.equal(sd["conv1.weight"], src["visual.conv1.weight"].half().float())
    assert torch.equal(sd["transformer.resblocks.0.attn.in_proj_bias"], src["visual.transformer.resblocks.0.attn.in_proj_bias"].half().float())
    assert torch.equal(sd["ln_pre.weight"], src["visual.ln_pre.weight"]) and torch.equal(sd["positional_embedding"], src["visual.positional_embedding"])
    assert all(v.dtype == torch.float32 for v in sd.values())
    assert len(net.state_dict()) == len(detgen.zutis_state_dict(cfg))       # the reference's key set (text embeddings are not parameters)


def test_update_text_embeddings_reloads_and_renormalises(monkeypatch, capsys):
    cfg = detgen.A4_TINY
    NZ, calls = _install(monkeypatch, cfg)
    net = NZ.ZUTIS(categories=["a", "b"], clip_arch="ViT-B/16", n_queries=cfg.n_queries, n_decoder_layers=cfg.dec_layers, n_heads=cfg.dec_heads,
                   device=torch.device("cpu"))
    before = net.text_embeddings.clone()
    del calls[:]
    net.update_text_embeddings(["zebra", "giraffe", "okapi"])               # zutis.py:333-338
    assert calls[0] == ("load", "ViT-B/16", "cpu") and calls[1] == ("tokenize", ("zebra", "giraffe", "okapi"))
    te = net.text_embeddings
    assert te.shape == (3, cfg.embed_dim) and te.dtype == torch.float32 and not te.requires_grad
    assert torch.allclose(te.norm(dim=1), torch.ones(3), atol=1e-6) and te.shape != before.shape
    assert "text embeddings have been changed for zebra, giraffe, okapi" in capsys.readouterr().out


def test_without_clip_the_constructor_says_what_to_pass(monkeypatch):
    import networks.zutis as NZ
    monkeypatch.setattr(NZ, "_clip", None)
    with pytest.raises(ImportError, match="text_embeddings="):
        NZ.ZUTIS(categories=["a"], clip_arch="ViT-B/16", device=torch.device("cpu"))
    net = NZ.ZUTIS(categories=["a"], clip_arch="ViT-B/32", device=torch.device("cpu"), n_decoder_layers=1,
                   text_embeddings=torch.from_numpy(detgen.text_embeddings(1, 512)))
    with pytest.raises(ImportError, match="update_text_embeddings needs"):
        net.update_text_embeddings(["b"])
