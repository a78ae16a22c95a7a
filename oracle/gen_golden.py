"""ORACLE tooling — generates tests/golden/*.npz by importing the REAL reference (authoring container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py
Needs /root/reference (read-only).  The reference never travels: only inputs' seeds (regenerated through
zutis_amd/detgen.py) and the reference's OUTPUTS are stored.  Stubs follow SURVEY.md Appendix B:
`clip` (random-init clip_arch.CLIP), `torchvision.ops.masks_to_boxes`, `pycocotools.mask.encode`, and the
SciPy>=1.14 `cg(tol->rtol)` shim.
"""
from __future__ import annotations

import os
import sys
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)

import numpy as np
import torch

from zutis_amd import detgen

GOLD = os.path.join(REPO, "tests", "golden")


def install_stubs(cfg: detgen.ZutisConfig):
    tv = types.ModuleType("torchvision")
    tvo = types.ModuleType("torchvision.ops")

    def masks_to_boxes(masks):
        out = torch.zeros((masks.shape[0], 4), dtype=torch.float32)
        for i, m in enumerate(masks):
            ys, xs = torch.where(m != 0)
            out[i] = torch.tensor([xs.min(), ys.min(), xs.max(), ys.max()], dtype=torch.float32)
        return out
    tvo.masks_to_boxes = masks_to_boxes
    tv.ops = tvo
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.ops"] = tvo
    pc = types.ModuleType("pycocotools")
    pcm = types.ModuleType("pycocotools.mask")
    pcm.encode = lambda m: {"size": list(m.shape), "mask": np.array(m)}   # stand-in: masks compared, not RLE bytes
    pc.mask = pcm
    sys.modules["pycocotools"] = pc
    sys.modules["pycocotools.mask"] = pcm
    clip = types.ModuleType("clip")

    def load(name, device=None):
        from networks.clip_arch import CLIP
        m = CLIP(embed_dim=cfg.embed_dim, image_resolution=cfg.patch * cfg.grid, vision_layers=cfg.layers,
                 vision_width=cfg.width, vision_patch_size=cfg.patch, context_length=8, vocab_size=64,
                 transformer_width=64, transformer_heads=1, transformer_layers=1)
        return m.float().eval(), None

    def tokenize(texts):
        t = torch.zeros((len(texts), 8), dtype=torch.long)
        t[:, 0] = 1
        t[:, -1] = 63
        return t
    clip.load, clip.tokenize = load, tokenize
    sys.modules["clip"] = clip


def build_reference_zutis(cfg: detgen.ZutisConfig, n_cat: int, seed: int = 1234):
    install_stubs(cfg)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for m in [k for k in sys.modules if k.startswith("networks") or k.startswith("utils")]:
        del sys.modules[m]
    from networks.zutis import ZUTIS
    net = ZUTIS(categories=[f"c{i}" for i in range(n_cat)], clip_arch="ViT-B/16", n_queries=cfg.n_queries,
                n_decoder_layers=cfg.dec_layers, n_heads=cfg.dec_heads, device=torch.device("cpu"))
    # decoder ff is fixed at 2048 by the reference ctor; rebuild layers' FFN only through state_dict shapes
    sd = {k: torch.from_numpy(v) for k, v in detgen.zutis_state_dict(cfg, seed).items()}
    net.load_state_dict(sd, strict=True)
    net.text_embeddings = torch.from_numpy(detgen.text_embeddings(n_cat, cfg.embed_dim))
    return net.eval().requires_grad_(False)


def gen_e2e(tag: str, cfg: detgen.ZutisConfig, b: int, H: int, W: int, n_cat: int, size, full: bool):
    net = build_reference_zutis(cfg, n_cat)
    x = torch.from_numpy(detgen.images(b, H, W))
    with torch.no_grad():
        enc_tokens, h, w = net.encoder(x)
        out = net(x)
        labels = net.predict(out, mask_type="semantic", size=size)
        logits_full = net.predict(out, mask_type="semantic", size=size, return_logits=True)
        logits_lo = net.predict(out, mask_type="semantic", size=None, return_logits=True)
    d = dict(b=b, H=H, W=W, n_cat=n_cat, size=np.array(size), labels=labels.astype(np.uint8 if n_cat < 256 else np.int16),
             logits_lo=logits_lo.numpy())
    if full:   # tiny config: store everything
        d.update(enc_tokens=enc_tokens.numpy(), mask_proposals=out["mask_proposals"].numpy(),
                 patch_tokens=out["patch_tokens"].numpy(), logits_full=logits_full.numpy())
        for nms in ("hard", "linear", "gaussian", None):
            with torch.no_grad():
                preds = net.predict(out, mask_type="instance", size=size, image_ids=list(range(b)), nms_type=nms)
            key = str(nms).lower()
            d[f"inst_{key}_n"] = len(preds)
            if len(preds):
                d[f"inst_{key}_masks"] = np.packbits(np.stack([p["segmentation"]["mask"] for p in preds]).astype(bool), axis=-1)
                d[f"inst_{key}_score"] = np.array([p["score"] for p in preds], np.float64)
                d[f"inst_{key}_cat"] = np.array([p["category_id"] for p in preds], np.int64)
                d[f"inst_{key}_img"] = np.array([p["image_id"] for p in preds], np.int64)
                d[f"inst_{key}_bbox"] = np.array([p["bbox"] for p in preds], np.float64)
    else:      # full-size: subsample the big tensors
        d.update(enc_tokens_sub=enc_tokens.numpy()[:, ::7, ::5], mask_proposals_sub=out["mask_proposals"].numpy()[:, :, ::9, ::3, ::3],
                 patch_tokens_sub=out["patch_tokens"].numpy()[:, ::3, ::3, ::4])
    np.savez_compressed(os.path.join(GOLD, f"e2e_{tag}.npz"), **d)
    print(tag, "labels hist", np.bincount(labels.reshape(-1))[:10], {k: getattr(v, "shape", v) for k, v in d.items()})


def gen_ops():
    """Per-op vectors from reference modules / the exact torch calls the reference makes."""
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import torch.nn.functional as F
    from networks.clip_arch import VisionTransformer
    from networks.positional_embedding import PositionEmbeddingSine
    from utils.running_score import RunningScore
    from utils.iou import compute_iou
    d = {}
    for g, (h, w) in [(14, (21, 21)), (14, (32, 32)), (7, (7, 7)), (14, (30, 40)), (4, (5, 7))]:
        pe = torch.from_numpy(detgen.det_normal(f"pe_{g}", (g * g + 1, 48)))
        d[f"posembed_g{g}_{h}x{w}"] = VisionTransformer.interpolate_positional_embedding(pe, (h, w))[0].numpy()
    for (h, w) in [(10, 14), (12, 17)]:
        pos = PositionEmbeddingSine(96 // 2 if h == 10 else 384, normalize=True)(torch.zeros(1, 1, h, w))
        d[f"sine_{h}x{w}"] = pos[0].permute(1, 2, 0).reshape(h * w, -1).numpy()
    x = torch.from_numpy(detgen.det_normal("up2", (2, 5, 7, 24)))
    d["up2"] = F.interpolate(x.permute(0, 3, 1, 2), scale_factor=2, mode="bilinear").permute(0, 2, 3, 1).numpy()
    lo = torch.from_numpy(detgen.det_normal("argmax_lo", (2, 9, 10, 14)))
    lo[0, 3] = lo[0, 5]                      # exact ties between classes 3 and 5 everywhere on image 0
    lo[1, :, 2:4] = 0.25                     # all-class ties on two rows of image 1
    d["argmax_lo"] = lo.numpy()
    d["argmax_labels_80x112"] = torch.argmax(F.interpolate(lo, size=(80, 112), mode="bilinear"), dim=1).numpy()
    d["argmax_labels_77x145"] = torch.argmax(F.interpolate(lo, size=(77, 145), mode="bilinear"), dim=1).numpy()
    rs = RunningScore(7)
    gt = (np.abs(detgen.det_normal("gt", (3, 20, 30))) * 3).astype(np.int64)
    gt[0, :2] = 255
    pr = (np.abs(detgen.det_normal("pr", (3, 20, 30))) * 3).astype(np.int64) % 7
    rs.update(gt, pr)
    sc, iu = rs.get_scores()
    d["rs_gt"], d["rs_pred"], d["rs_hist"] = gt, pr, rs.confusion_matrix
    d["rs_scores"] = np.array([sc["Pixel Acc"], sc["Mean Acc"], sc["FreqW Acc"], sc["Mean IoU"]])
    m1 = detgen.det_normal("m1", (16, 16)) > 0
    m2 = detgen.det_normal("m2", (16, 16)) > 0.3
    d["iou_m1"], d["iou_m2"], d["iou"] = m1, m2, np.array(compute_iou(m1, m2))
    np.savez_compressed(os.path.join(GOLD, "ops.npz"), **d)
    print("ops:", {k: v.shape for k, v in d.items()})


def gen_selfmask():
    """SelfMask reference (random init + deterministic weights), train-mode dict and inference masks."""
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for m in [k for k in sys.modules if k.startswith("networks") or k.startswith("utils")]:
        del sys.modules[m]
    from networks.selfmask.selfmask import SelfMask
    net = SelfMask()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in detgen.selfmask_state_dict().items()}, strict=True)
    net.eval().requires_grad_(False)
    d = {}
    for tag, (b, H, W) in {"small": (2, 72, 100), "full": (1, 224, 300)}.items():
        x = torch.from_numpy(detgen.images(b, H, W, seed=11))
        with torch.no_grad():
            o = net(x)
            inf = net(x, inference=True, bilateral_solver=False)
            # encoder_only=True is broken in the reference (selfmask.py:162 views a non-contiguous [b,D,hw] as [b,h,w,D])
        d[f"{tag}_shape"] = np.array([b, H, W])
        d[f"{tag}_objectness"] = o["objectness"].numpy()
        d[f"{tag}_mask_pred"] = o["mask_pred"].numpy() if tag == "small" else o["mask_pred"].numpy()[:, :, :, ::2, ::2]
        d[f"{tag}_dts"] = np.packbits(np.stack([t.numpy() for t in inf["dts"]]).astype(bool), axis=-1)
        print("selfmask", tag, {k: v.shape for k, v in d.items() if k.startswith(tag)}, "fg frac",
              float(np.mean([t.float().mean() for t in inf["dts"]])))
    np.savez_compressed(os.path.join(GOLD, "selfmask.npz"), **d)


def gen_bilateral():
    """Reference bilateral solver (utils/bilateral_solver.py) on small synthetic images + the gray-ramp known answer."""
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for m in [k for k in sys.modules if k.startswith("networks") or k.startswith("utils")]:
        del sys.modules[m]
    import scipy.sparse.linalg as spla
    import utils.bilateral_solver as rb
    from PIL import Image
    its = []

    def cg_shim(A, b, x0=None, M=None, maxiter=None, tol=1e-5):
        n_it = [0]
        x, info = spla.cg(A, b, x0=x0, M=M, maxiter=maxiter, rtol=tol, atol=0.0, callback=lambda xk: n_it.__setitem__(0, n_it[0] + 1))
        its.append(n_it[0])
        return x, info
    rb.cg = cg_shim
    d = {}
    ramp = np.repeat(np.arange(0, 256, 16, dtype=np.uint8)[None, :, None], 3, axis=2)       # gray 0,16,...,240
    g = rb.BilateralGrid(np.repeat(ramp, 4, axis=0), sigma_spatial=16, sigma_luma=16, sigma_chroma=8)
    yuv = rb.rgb2yuv(ramp)
    d["ramp_luma_bins"] = (yuv[0, :, 0] / 16).astype(int)
    d["ramp_chroma_bins"] = (yuv[0, :, 1:] / 8).astype(int)
    for tag, (h, w, seed) in {"a": (96, 128, 3), "b": (120, 90, 5), "c": (64, 64, 9)}.items():
        rgb = detgen.selfmask_like_rgb(h, w, seed=seed)
        yy, xx = np.mgrid[:h, :w]
        target = (((yy - h * 0.5) ** 2 + (xx - w * 0.45) ** 2) < (0.3 * min(h, w)) ** 2).astype(np.uint8)
        if tag == "c":
            target[:8, :8] = 1                                                             # a second component
        grid = rb.BilateralGrid(rgb, sigma_spatial=16, sigma_luma=16, sigma_chroma=8)
        Dn, Dm = rb.bistochastize(grid)
        soft, binary = rb.bilateral_solver_output(Image.fromarray(rgb), target)
        d[f"{tag}_hw"] = np.array([h, w, seed])
        d[f"{tag}_target"] = target
        d[f"{tag}_nvertices"] = grid.nvertices
        d[f"{tag}_nnz"] = np.array([b.nnz for b in grid.blurs])
        d[f"{tag}_n"], d[f"{tag}_m"] = Dn.diagonal(), Dm.diagonal()
        d[f"{tag}_soft"], d[f"{tag}_binary"] = soft, binary
        d[f"{tag}_cg_iters"] = its[-1]
        print("bilateral", tag, "V", grid.nvertices, "nnz", d[f"{tag}_nnz"], "cg iters", its[-1], "fg", (soft > 0.5).mean())
    # EMPTY target (no foreground found by the pseudo-labeller): b = splat(0) = 0, and scipy's cg returns zeros without iterating
    # (`if bnrm2 == 0: return postprocess(b), 0`); the post-processing then finds no component and falls back to the all-True mask
    # (bilateral_solver.py:188-193)
    rgb = detgen.selfmask_like_rgb(96, 128, seed=3)
    soft, binary = rb.bilateral_solver_output(Image.fromarray(rgb), np.zeros((96, 128), np.uint8))
    d["z_soft"], d["z_binary"], d["z_cg_iters"] = soft, binary, its[-1]
    print("bilateral empty target: soft max", float(np.abs(soft).max()), "finite", bool(np.isfinite(soft).all()), "binary all-True", bool(binary.all()), "cg iters", its[-1])
    # de-normalise step (utils/utils.py:261-273) on values engineered to land on integer boundaries
    from utils.utils import convert_tensor_to_pil_image
    x = torch.from_numpy(detgen.det_normal("denorm", (3, 40, 56)))
    x[0, 0, :16] = torch.from_numpy(((np.arange(16) * 16 / 255.0 - 0.485) / 0.229).astype(np.float32))
    d["denorm_u8"] = np.array(convert_tensor_to_pil_image(x))
    np.savez_compressed(os.path.join(GOLD, "bilateral.npz"), **d)


def gen_text():
    """CLIP.encode_text (clip_arch.py:534-547) of the REAL reference class + the prompt-ensembling loop of
    utils/extract_text_embeddings.py:98-115 (run through the reference function with a stub tokenizer)."""
    install_stubs(detgen.TINY)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from networks.clip_arch import CLIP
    d = {}
    for tag, tc, n in (("tiny", detgen.TEXT_TINY, 9), ("b", detgen.TEXT_B, 6)):
        m = CLIP(embed_dim=tc.embed_dim, image_resolution=32, vision_layers=1, vision_width=64, vision_patch_size=16,
                 context_length=tc.context_length, vocab_size=tc.vocab_size, transformer_width=tc.width,
                 transformer_heads=tc.heads, transformer_layers=tc.layers).float().eval()
        sd = {k: torch.from_numpy(v) for k, v in detgen.clip_text_state_dict(tc).items()}
        missing, unexpected = m.load_state_dict(sd, strict=False)
        assert not unexpected and all(k.startswith("visual.") or k == "logit_scale" for k in missing), (missing, unexpected)
        tok = torch.from_numpy(detgen.text_tokens(n, tc))
        with torch.no_grad():
            d[f"{tag}_encode_text"] = m.encode_text(tok).numpy()
        if tag == "tiny":
            # the reference's own ensembling loop, with clip.tokenize replaced by our deterministic token rows
            import clip as clip_stub
            import utils.extract_text_embeddings as ete
            C, T = 3, 5
            toks = torch.from_numpy(detgen.text_tokens(C * T, tc, seed=23)).view(C, T, -1)
            table = {f"cat{c}": toks[c] for c in range(C)}
            clip_stub.tokenize = lambda texts: _FakeCuda(table[texts[0].split("|")[1]])
            ete.clip = clip_stub
            ete.tqdm = lambda x: x
            res = ete.extract_text_embeddings(m, [f"cat{c}" for c in range(C)], [f"t{t}|{{}}|" for t in range(T)])
            d["tiny_prompt_ensemble"] = np.stack([res[f"cat{c}"].numpy() for c in range(C)])
    np.savez_compressed(os.path.join(GOLD, "text.npz"), **d)
    print("text.npz", {k: v.shape for k, v in d.items()})


def gen_c3():
    """Config 3 at its own shape (coco20k_eval.py:241-268): ViT-B/16, batch 1, native-resolution COCO-like inputs, forward +
    predict(mask_type="instance", size=image size, nms_type="hard" | "linear" | None).  Stores the per-prediction integers /
    scores / boxes, the masks bit-packed, and sub-sampled forward tensors.
    Weights = detgen.c3_state_dict (queries that differ: at plain random init hard NMS leaves one survivor of one class),
    threshold = detgen.C3_THRESHOLD, and text embeddings made orthogonal to the mean patch token of the first image (stored in
    the fixture as an INPUT: with random text rows every region's average token picks the same class) -> ~9 categories and
    12-17 hard-NMS survivors out of ~100 candidates per image, ids that wrap CPython's set table included."""
    cfg = detgen.VIT_B16
    net = build_reference_zutis(cfg, 81)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in detgen.c3_state_dict(cfg).items()}, strict=True)
    thr = detgen.C3_THRESHOLD
    d = {}
    for (H, W) in ((480, 640), (427, 640)):
        tag = f"{H}x{W}"
        x = torch.from_numpy(detgen.images(1, H, W, seed=21))
        with torch.no_grad():
            out = net(x)
            if "text" not in d:
                t0 = torch.from_numpy(detgen.text_embeddings(81, cfg.embed_dim))
                mu = out["patch_tokens"].reshape(-1, cfg.embed_dim).mean(0)
                mu = mu / mu.norm()
                t = t0 - (t0 @ mu)[:, None] * mu[None]
                d["text"] = (t / t.norm(dim=1, keepdim=True)).numpy()
                net.text_embeddings = torch.from_numpy(d["text"])
            logits_lo = net.predict(out, mask_type="semantic", size=None, return_logits=True)
        d[f"{tag}_mask_proposals_last_sub"] = out["mask_proposals"].numpy()[:, -1, :, ::3, ::3]
        d[f"{tag}_patch_tokens_sub"] = out["patch_tokens"].numpy()[:, ::3, ::3, ::4]
        d[f"{tag}_logits_lo_sub"] = logits_lo.numpy()[:, :, ::2, ::2]
        for nms, key in (("hard", ""), ("linear", "linear_")):
            with torch.no_grad():
                preds = net.predict(out, mask_type="instance", threshold=thr, size=(H, W), image_ids=[7], nms_type=nms)
            d[f"{tag}_{key}n"] = len(preds)
            d[f"{tag}_{key}score"] = np.array([p["score"] for p in preds], np.float64)
            d[f"{tag}_{key}cat"] = np.array([p["category_id"] for p in preds], np.int64)
            d[f"{tag}_{key}area"] = np.array([int(p["segmentation"]["mask"].sum()) for p in preds], np.int64)
            if nms == "hard":
                d[f"{tag}_masks"] = np.packbits(np.stack([p["segmentation"]["mask"] for p in preds]).astype(bool), axis=-1)
                d[f"{tag}_bbox"] = np.array([p["bbox"] for p in preds], np.float64)
            print("c3", tag, nms, "preds", len(preds), "cats in emission order", [p["category_id"] for p in preds])
        with torch.no_grad():     # every candidate (no NMS): per-query class / score / area / box at the native resolution
            allp = net.predict(out, mask_type="instance", threshold=thr, size=(H, W), image_ids=[7], nms_type=None)
        d[f"{tag}_all_n"] = len(allp)
        d[f"{tag}_all_score"] = np.array([p["score"] for p in allp], np.float64)
        d[f"{tag}_all_cat"] = np.array([p["category_id"] for p in allp], np.int64)
        d[f"{tag}_all_bbox"] = np.array([p["bbox"] for p in allp], np.float64)
        d[f"{tag}_all_area"] = np.array([int(p["segmentation"]["mask"].sum()) for p in allp], np.int64)
        print("c3", tag, "no-NMS candidates", len(allp), "categories", sorted(set(int(c) for c in d[f"{tag}_all_cat"])))
    np.savez_compressed(os.path.join(GOLD, "c3_vitb16.npz"), **d)


A4_CFG = detgen.A4_TINY


def gen_a4():
    """build_model / convert_weights (clip_arch.py:566-627) through the real ZUTIS constructor (zutis.py:35-55): `clip.load`
    hands back a CLIP carrying GENERIC fp32 weights (detgen.clip_full_state_dict); the constructor re-builds the model from
    its state_dict, which rounds conv / Linear / attention / proj weights (and Linear biases) through fp16 and leaves
    LayerNorm / class / positional embeddings in fp32, then casts back to fp32.  Stores the inferred architecture, per-key
    fingerprints of the resulting encoder parameters and the forward outputs on a fixed input."""
    cfg = A4_CFG
    install_stubs(cfg)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for m in [k for k in sys.modules if k.startswith("networks") or k.startswith("utils")]:
        del sys.modules[m]
    import clip as clip_stub
    from networks.clip_arch import CLIP
    csd = {k: torch.from_numpy(v) for k, v in detgen.clip_full_state_dict(cfg).items()}

    def load(name, device=None):
        m = CLIP(embed_dim=cfg.embed_dim, image_resolution=cfg.patch * cfg.grid, vision_layers=cfg.layers, vision_width=cfg.width,
                 vision_patch_size=cfg.patch, context_length=8, vocab_size=64, transformer_width=64, transformer_heads=1,
                 transformer_layers=1).float()
        missing, unexpected = m.load_state_dict(csd, strict=False)
        assert not unexpected and set(missing) <= {"logit_scale"}, (missing, unexpected)
        return m.eval(), None
    clip_stub.load = load
    from networks.zutis import ZUTIS
    net = ZUTIS(categories=[f"c{i}" for i in range(7)], clip_arch="ViT-B/16", n_queries=cfg.n_queries,
                n_decoder_layers=cfg.dec_layers, n_heads=cfg.dec_heads, device=torch.device("cpu"))
    head = {k: torch.from_numpy(v) for k, v in detgen.zutis_state_dict(cfg).items() if not k.startswith("encoder.")}
    missing, unexpected = net.load_state_dict(head, strict=False)
    assert not unexpected and all(k.startswith("encoder.") for k in missing)
    net.text_embeddings = torch.from_numpy(detgen.text_embeddings(7, cfg.embed_dim))
    net.eval().requires_grad_(False)
    enc = net.encoder
    d = {"arch": np.array([enc.width, enc.transformer.layers, enc.conv1.kernel_size[0], enc.input_resolution, enc.output_dim])}
    for k, v in enc.state_dict().items():
        a = v.numpy().astype(np.float64).reshape(-1)
        d["fp_" + k] = np.concatenate([[a.sum(), np.abs(a).sum(), (a * a).sum()], a[:8], [float(v.dtype == torch.float32)]])
    x = torch.from_numpy(detgen.images(2, 80, 112))
    with torch.no_grad():
        tok, h, w = enc(x)
        out = net(x)
    d["enc_tokens"], d["mask_proposals"], d["patch_tokens"] = tok.numpy(), out["mask_proposals"].numpy(), out["patch_tokens"].numpy()
    np.savez_compressed(os.path.join(GOLD, "a4_build_model.npz"), **d)
    print("a4", d["arch"], {k: v.shape for k, v in d.items() if not k.startswith("fp_")}, len([k for k in d if k.startswith("fp_")]), "keys")


E1_CASES = {"small": (detgen.ZutisConfig(width=128, layers=2, patch=14, grid=3, embed_dim=64), 3),
            "l14_336": (detgen.ZutisConfig(width=1024, layers=2, patch=14, grid=24, embed_dim=768), 2)}   # ViT-L/14@336 geometry, 2 layers


def gen_encode_image():
    """E1 (utils/extract_image_embeddings.py:72-73): `model.encode_image(x)` = `self.visual(x)` (clip_arch.py:531-532) followed by
    the L2 normalisation.  The third-party `clip` package is absent, but the reference's own VisionTransformer keeps CLIP's
    ORIGINAL forward as a comment (clip_arch.py:413-431) next to the modified one — so the real submodules (conv1 / class_embedding /
    positional_embedding / ln_pre / transformer / ln_post / proj, clip_arch.py:335-354) are run here in exactly that order.
    Stores only the outputs: unit-norm embeddings for a small tower and for the ViT-L/14@336 geometry (24x24 grid, D = 1024,
    2 layers); inputs / weights are regenerated from zutis_amd/detgen.py by the tests."""
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for m in [k for k in sys.modules if k.startswith("networks") or k.startswith("utils")]:
        del sys.modules[m]
    from networks.clip_arch import VisionTransformer
    d = {}
    for tag, (cfg, B) in E1_CASES.items():
        R = cfg.patch * cfg.grid
        vt = VisionTransformer(input_resolution=R, patch_size=cfg.patch, width=cfg.width, layers=cfg.layers, heads=cfg.width // 64,
                               output_dim=cfg.embed_dim)
        sd = {k[len("encoder."):]: torch.from_numpy(v) for k, v in detgen.zutis_state_dict(cfg).items() if k.startswith("encoder.")}
        vt.load_state_dict(sd, strict=True)
        vt.eval().requires_grad_(False)
        x = torch.from_numpy(detgen.images(B, R, R, seed=5))
        with torch.no_grad():
            t = vt.conv1(x)
            t = t.reshape(t.shape[0], t.shape[1], -1).permute(0, 2, 1)
            t = torch.cat([vt.class_embedding.to(t.dtype) + torch.zeros(t.shape[0], 1, t.shape[-1], dtype=t.dtype), t], dim=1)
            t = t + vt.positional_embedding.to(t.dtype)
            t = vt.ln_pre(t)
            t = vt.transformer(t.permute(1, 0, 2)).permute(1, 0, 2)
            e = vt.ln_post(t[:, 0, :]) @ vt.proj
            e = e / torch.linalg.norm(e, ord=2, dim=1, keepdim=True)               # extract_image_embeddings.py:73
        d[f"{tag}_shape"] = np.array([B, R, cfg.width, cfg.layers, cfg.patch, cfg.grid, cfg.embed_dim])
        d[f"{tag}_embeddings"] = e.numpy()
        print("encode_image", tag, e.shape, float(e.norm(dim=1).mean()))
    np.savez_compressed(os.path.join(GOLD, "encode_image.npz"), **d)


class _FakeCuda:
    """`clip.tokenize(texts).cuda()` in the reference loop: hand the CPU tensor back."""
    def __init__(self, t):
        self.t = t

    def cuda(self):
        return self.t


if __name__ == "__main__":
    assert os.path.isdir(REF), "reference not mounted"
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    if "--bilateral-only" in sys.argv:
        gen_bilateral()
        sys.exit(0)
    if "--text-only" in sys.argv:
        gen_text()
        sys.exit(0)
    if "--c3-only" in sys.argv:
        gen_c3()
        sys.exit(0)
    if "--a4-only" in sys.argv:
        gen_a4()
        sys.exit(0)
    if "--selfmask-only" in sys.argv:
        gen_selfmask()
        sys.exit(0)
    if "--encode-image-only" in sys.argv:
        gen_encode_image()
        sys.exit(0)
    gen_ops()
    gen_selfmask()
    gen_bilateral()
    gen_text()
    gen_e2e("tiny", detgen.TINY, b=2, H=80, W=112, n_cat=7, size=(80, 112), full=True)
    gen_e2e("vitb16_336", detgen.VIT_B16, b=1, H=336, W=336, n_cat=81, size=(336, 336), full=False)
    gen_e2e("vitb32_224", detgen.VIT_B32, b=1, H=224, W=224, n_cat=81, size=(224, 224), full=False)
    gen_c3()
    gen_a4()
    gen_encode_image()
