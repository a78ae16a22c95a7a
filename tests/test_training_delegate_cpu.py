"""The overlay's autograd story (SURVEY 8b "fall back cleanly"; trainer.py:136 back-propagates through ZUTIS.forward):

* the reference's own networks/zutis.py importable beside the overlay (zutis_amd.dropin.networks.zutis.REFERENCE_MODULE_NAMES) ->
  forward() under autograd DELEGATES to it, over the overlay's own Parameter objects;
* not importable -> NotImplementedError (the HIP path is inference-only);
* no_grad / frozen parameters never take that route (the engine is asked, and on this CPU-only box refuses to run).

CPU only.  The first test uses a stand-in "reference" class; the second the REAL reference when /root/reference is present (this
authoring container; skipped on the GPU box, where it does not exist)."""
import os
import sys
import types

import pytest
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(REPO, "zutis_amd", "dropin")
sys.path.insert(0, DROPIN)
from zutis_amd import detgen  # noqa: E402

REF = "/root/reference"


def _overlay(monkeypatch, cfg):
    """The overlay module with a stand-in `clip` (the constructor's clip.load branch), as tests/test_dropin_clip_cpu.py installs it."""
    import networks.zutis as NZ
    if not hasattr(NZ, "reference_zutis_class"):           # another test left the reference's module under this name
        for m in [k for k in sys.modules if k == "networks" or k.startswith("networks.")]:
            del sys.modules[m]
        import networks.zutis as NZ
    stub = types.ModuleType("clip")
    sd = {k: torch.from_numpy(v) for k, v in detgen.clip_full_state_dict(cfg).items()}

    class _M:
        def encode_text(self, tokens):
            return torch.from_numpy(detgen.det_normal("dlg.text", (tokens.shape[0], cfg.embed_dim), 1.0, 0.0, 3)).to(torch.float16)

        def state_dict(self):
            return sd
    stub.load = lambda name, device=None: (_M(), None)
    stub.tokenize = lambda texts: torch.zeros((len(texts), 8), dtype=torch.long)
    monkeypatch.setattr(NZ, "_clip", stub)
    return NZ


def _make(NZ, cfg):
    net = NZ.ZUTIS(categories=["a", "b", "c"], clip_arch="ViT-B/16", n_queries=cfg.n_queries, n_decoder_layers=cfg.dec_layers,
                   n_heads=cfg.dec_heads, device=torch.device("cpu"))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in detgen.zutis_state_dict(cfg).items()}, strict=True)
    return net


def test_refuses_without_a_reference_and_delegates_with_one(monkeypatch):
    cfg = detgen.A4_TINY
    NZ = _overlay(monkeypatch, cfg)
    for name in NZ.REFERENCE_MODULE_NAMES:
        monkeypatch.delitem(sys.modules, name, raising=False)
    monkeypatch.delenv("ZUTIS_REFERENCE_MODULE", raising=False)
    net = _make(NZ, cfg).train()
    x = torch.from_numpy(detgen.images(1, 4 * cfg.patch, 5 * cfg.patch))
    assert NZ.reference_zutis_class() is None
    with pytest.raises(NotImplementedError, match="training delegate"):
        net(x)                                             # autograd on, parameters trainable, no reference: refused, loudly

    built = []

    class FakeReference(nn.Module):
        """Same constructor signature and parameter names as the reference's ZUTIS (the overlay's own containers provide them);
        forward = a differentiable function of two parameters, enough to see gradients arrive in the OVERLAY's tensors."""

        def __init__(self, **kw):
            super().__init__()
            built.append(kw)
            inner = NZ.ZUTIS(**kw)
            for n, m in inner.named_children():
                self.add_module(n, m)
            self.query_embed = inner.query_embed
            self.text_embeddings = None

        def forward(self, x):
            s = x.mean() * self.query_embed.sum() + self.encoder.proj.sum()
            return {"mask_proposals": s.reshape(1), "patch_tokens": s.reshape(1), "mode": self.training}
    mod = types.ModuleType("zutis_reference_networks_zutis")
    mod.ZUTIS = FakeReference
    monkeypatch.setitem(sys.modules, "zutis_reference_networks_zutis", mod)
    assert NZ.reference_zutis_class() is FakeReference

    out = net(x)
    assert out["mode"] is True and len(built) == 1
    assert built[0]["categories"] == ["a", "b", "c"] and built[0]["n_queries"] == cfg.n_queries and built[0]["clip_arch"] == "ViT-B/16"
    d = net._delegate
    mine, theirs = dict(net.named_parameters()), dict(d.named_parameters())
    assert set(mine) == set(theirs) and all(theirs[k] is mine[k] for k in mine)      # ONE set of Parameter objects
    assert d.text_embeddings is net.text_embeddings
    assert "_delegate" not in dict(net.named_modules()) and len(net.state_dict()) == len(detgen.zutis_state_dict(cfg))
    # gradients land in the overlay's parameters; an optimizer over net.parameters() (main.py) moves what the HIP engine will pack
    eng = net._get_engine()
    key0 = eng._version_key()
    out["mask_proposals"].sum().backward()
    assert net.query_embed.grad is not None and torch.all(net.query_embed.grad == x.mean())
    assert net.encoder.proj.grad is not None and net.ffn1.layers[0].weight.grad is None
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    before = net.query_embed.detach().clone()
    opt.step()
    assert not torch.equal(net.query_embed.detach(), before)
    assert eng._version_key() != key0                      # the engine re-packs on its next forward
    # eval() is mirrored; inference never delegates: no_grad goes to the HIP engine, which has no GPU here and says so
    net.eval()
    assert net(x)["mode"] is False                         # grad still enabled + trainable parameters: still the delegate
    from zutis_amd._lib import ZutisHipError
    with torch.no_grad():
        with pytest.raises((ZutisHipError, RuntimeError, AssertionError)):
            net(x)
    net.requires_grad_(False)
    with pytest.raises((ZutisHipError, RuntimeError, AssertionError)):
        net(x)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "networks")), reason="needs the reference checkout (authoring container only)")
def test_delegate_is_the_real_reference(monkeypatch):
    """With the one-line arrangement of INTEGRATION.md ("Training") the overlay's forward under autograd returns the REAL reference
    module's outputs (bitwise: it is the reference's code over the same parameters), back-propagates, and state_dict round-trips."""
    sys.path.insert(0, REPO)
    from oracle import gen_golden as G
    cfg = detgen.TINY
    saved = {k: v for k, v in sys.modules.items() if k == "networks" or k.startswith("networks.") or k == "utils" or k.startswith("utils.")
             or k in ("clip", "torchvision", "torchvision.ops", "pycocotools", "pycocotools.mask")}
    path0 = list(sys.path)
    try:
        G.install_stubs(cfg)                               # stand-ins for clip / torchvision.ops / pycocotools
        for m in [k for k in sys.modules if k == "networks" or k.startswith("networks.") or k == "utils" or k.startswith("utils.")]:
            del sys.modules[m]
        sys.path.insert(0, REF)
        import networks.zutis as ref_mod                   # the reference's own module ...
        import networks.clip_arch as ref_arch
        assert ref_mod.__file__.startswith(REF)

        def load(name, device=None):                       # the stand-in `clip.load`: the reference's own CLIP class, random init
            m = ref_arch.CLIP(embed_dim=cfg.embed_dim, image_resolution=cfg.patch * cfg.grid, vision_layers=cfg.layers, vision_width=cfg.width,
                              vision_patch_size=cfg.patch, context_length=8, vocab_size=64, transformer_width=64, transformer_heads=1,
                              transformer_layers=1)
            return m.float().eval(), None
        sys.modules["clip"].load = load                    # (gen_golden's stub imports networks.clip_arch by name: shadowed below)
        sys.modules["zutis_reference_networks_zutis"] = ref_mod      # ... kept importable under another name: THE one line
        for m in [k for k in sys.modules if k == "networks" or k.startswith("networks.")]:
            del sys.modules[m]
        sys.path.remove(REF)
        sys.path.insert(0, DROPIN)
        import networks.zutis as NZ                        # the overlay shadows networks.zutis from here on
        assert NZ.__file__.startswith(DROPIN) and NZ.reference_zutis_class() is ref_mod.ZUTIS
        sys.path.insert(1, REF)                            # the reference's constructor imports its own utils.* / networks.* siblings lazily
        monkeypatch.setattr(NZ, "_clip", sys.modules["clip"])
        net = NZ.ZUTIS(categories=[f"c{i}" for i in range(7)], clip_arch="ViT-B/16", n_queries=cfg.n_queries,
                       n_decoder_layers=cfg.dec_layers, n_heads=cfg.dec_heads, device=torch.device("cpu"))
        sd = {k: torch.from_numpy(v) for k, v in detgen.zutis_state_dict(cfg).items()}
        net.load_state_dict(sd, strict=True)
        net.train()
        x = torch.from_numpy(detgen.images(2, 80, 112))
        out = net(x)                                       # autograd on -> the reference's forward over the overlay's parameters
        assert isinstance(net._delegate, ref_mod.ZUTIS)
        # the reference built standalone with the same weights answers the same, bit for bit
        ref = ref_mod.ZUTIS(categories=[f"c{i}" for i in range(7)], clip_arch="ViT-B/16", n_queries=cfg.n_queries,
                            n_decoder_layers=cfg.dec_layers, n_heads=cfg.dec_heads, device=torch.device("cpu"))
        ref.load_state_dict(sd, strict=True)
        ref.train()
        want = ref(x)
        assert torch.equal(out["mask_proposals"], want["mask_proposals"]) and torch.equal(out["patch_tokens"], want["patch_tokens"])
        loss = out["mask_proposals"].mean() + out["patch_tokens"].square().mean()
        loss.backward()
        g = net.decoder.layers[0].linear1.weight.grad
        assert g is not None and torch.isfinite(g).all() and float(g.abs().max()) > 0
        assert net.encoder.conv1.weight.grad is not None
        assert set(net.state_dict()) == set(sd)
    finally:
        sys.path[:] = path0
        for m in [k for k in sys.modules if k == "networks" or k.startswith("networks.") or k == "utils" or k.startswith("utils.")
                  or k in ("clip", "torchvision", "torchvision.ops", "pycocotools", "pycocotools.mask", "zutis_reference_networks_zutis")]:
            del sys.modules[m]
        sys.modules.update(saved)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "networks")), reason="needs the reference checkout (authoring container only)")
def test_keep_reference_helper_leaves_the_overlay_canonical(monkeypatch):
    """`keep_reference_for_training(<reference>/networks/zutis.py)` (round-5 advisor: the hand-written alias left the reference's
    `networks` / `utils` cached and the overlay was silently never loaded): the reference's class becomes the delegate, and afterwards
    `networks.zutis`, `utils.running_score`, ... are still the overlay's modules — a later `from networks.zutis import ZUTIS`
    (utils/utils.py:166-174) gets the HIP module."""
    sys.path.insert(0, REPO)
    from oracle import gen_golden as G
    cfg = detgen.TINY
    keep = ("clip", "torchvision", "torchvision.ops", "pycocotools", "pycocotools.mask")
    saved = {k: v for k, v in sys.modules.items() if k == "networks" or k.startswith("networks.") or k == "utils" or k.startswith("utils.") or k in keep}
    path0 = list(sys.path)
    try:
        G.install_stubs(cfg)
        for m in [k for k in sys.modules if k == "networks" or k.startswith("networks.") or k == "utils" or k.startswith("utils.")]:
            del sys.modules[m]
        sys.modules.pop("zutis_reference_networks_zutis", None)
        sys.path.insert(0, DROPIN)
        import networks.zutis as NZ
        import utils.running_score as RS
        assert NZ.__file__.startswith(DROPIN) and RS.__file__.startswith(DROPIN) and NZ.reference_zutis_class() is None
        cls = NZ.keep_reference_for_training(os.path.join(REF, "networks", "zutis.py"))
        assert cls.__module__ == "zutis_reference_networks_zutis" and NZ.reference_zutis_class() is cls and cls is not NZ.ZUTIS
        import importlib
        assert importlib.import_module("networks.zutis") is NZ and sys.modules["networks"].__file__.startswith(DROPIN)
        assert importlib.import_module("utils.running_score") is RS and sys.modules["utils"].__file__.startswith(DROPIN)
        from networks.zutis import ZUTIS as again
        assert again is NZ.ZUTIS
        with pytest.raises(ImportError):
            NZ.keep_reference_for_training(NZ.__file__)            # the overlay's own file is not a reference
        assert NZ.reference_zutis_class() is None or NZ.reference_zutis_class() is not NZ.ZUTIS
    finally:
        sys.path[:] = path0
        for m in [k for k in sys.modules if k == "networks" or k.startswith("networks.") or k == "utils" or k.startswith("utils.")
                  or k in keep + ("zutis_reference_networks_zutis",)]:
            del sys.modules[m]
        sys.modules.update(saved)


def test_delegate_follows_update_text_embeddings(monkeypatch):
    """The delegate's text_embeddings is re-bound on every call (it was bound once and went stale after update_text_embeddings)."""
    cfg = detgen.A4_TINY
    NZ = _overlay(monkeypatch, cfg)

    class Fake(nn.Module):
        def __init__(self, **kw):
            super().__init__()
            inner = NZ.ZUTIS(**kw)
            for n, m in inner.named_children():
                self.add_module(n, m)
            self.query_embed = inner.query_embed
            self.text_embeddings = None

        def forward(self, x):
            return {"te": self.text_embeddings}
    mod = types.ModuleType("zutis_reference_networks_zutis")
    mod.ZUTIS = Fake
    monkeypatch.setitem(sys.modules, "zutis_reference_networks_zutis", mod)
    net = _make(NZ, cfg).train()
    x = torch.from_numpy(detgen.images(1, 4 * cfg.patch, 5 * cfg.patch))
    assert net(x)["te"] is net.text_embeddings
    net.text_embeddings = torch.ones((2, cfg.embed_dim))          # what update_text_embeddings() does (zutis.py:333-338)
    assert net(x)["te"] is net.text_embeddings
