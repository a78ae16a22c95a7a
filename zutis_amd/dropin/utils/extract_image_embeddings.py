"""Drop-in replacement for the reference's utils/extract_image_embeddings.py on MI355X.

extract_image_embeddings(p_images, model_name, fp, device, batch_size, n_workers) -> Dict[basename, FloatTensor[E]]
keeps the reference signature and on-disk pickle format (utils/extract_image_embeddings.py:21-86).  The reference runs
third-party `clip`'s fp16 `encode_image` (`clip.load` leaves the model in half precision on a GPU, :43,72-76); here the ViT
tower runs in zutis_amd.engine.ClipImageEncoder.  `precision="exact"` (the default) computes every contraction in the fp32-class
f16x3 / f16x2 mode (fp16 split pairs, fp32 accumulate / residual / LayerNorm) — MORE precise than the reference's own run;
`precision="fast"` is the reference's arithmetic class for this path (fp16 MFMA operands in the transformer body, fp32 accumulate).  Weights come from `clip.load` when the package is present, else from `state_dict=`.
Pre-processing (bicubic resize of the shorter side, centre crop, CLIP mean/std; SimpleDataset :90-116) is PIL + NumPy
on the host — data loading is outside the hot path.
"""
import os
import pickle as pkl
from typing import Dict, List, Optional

import numpy as np
import torch
from PIL import Image

from zutis_amd.engine import ClipImageEncoder

_MEAN = np.array((0.48145466, 0.4578275, 0.40821073), np.float32)
_STD = np.array((0.26862954, 0.26130258, 0.27577711), np.float32)


def resize_crop_box(w: int, h: int, n_px: int):
    """CLIP's `_transform` (third-party clip, used at utils/extract_image_embeddings.py:43,98-103) =
    torchvision Resize(n_px, BICUBIC) + CenterCrop(n_px), with torchvision's integer conventions: the shorter side becomes
    n_px and the longer one int(n_px * long / short) (truncation); the crop offset is int(round((side - n_px) / 2.0))
    (Python's round-half-to-even).  Returns ((new_w, new_h), (left, top))."""
    if w <= h:
        nw, nh = n_px, int(n_px * h / w)
    else:
        nw, nh = int(n_px * w / h), n_px
    return (nw, nh), (int(round((nw - n_px) / 2.0)), int(round((nh - n_px) / 2.0)))


def _preprocess(p_image: str, n_px: int) -> np.ndarray:
    im = Image.open(p_image).convert("RGB")
    (nw, nh), (left, top) = resize_crop_box(*im.size, n_px)
    im = im.resize((nw, nh), Image.BICUBIC)
    a = np.asarray(im.crop((left, top, left + n_px, top + n_px)), np.float32) / 255.0
    return ((a - _MEAN) / _STD).transpose(2, 0, 1)


@torch.no_grad()
def extract_image_embeddings(
        p_images: List[str],
        model_name: str = "RN50",
        fp: Optional[str] = None,
        device: torch.device = torch.device("cuda:0"),
        batch_size: int = 256,
        n_workers: int = 16,
        *,
        state_dict: Optional[dict] = None,
        precision: str = "exact",
) -> Dict[str, torch.Tensor]:
    if "ViT" not in model_name:
        raise NotImplementedError(f"{model_name}: only the CLIP-ViT towers are on the MI355X hot path")
    if state_dict is None:
        import clip                                                    # reference :43 (needs the package + weights)
        model, _ = clip.load(model_name, device="cpu")
        state_dict = model.state_dict()
    vis = {k: v.float().to(device) for k, v in state_dict.items() if k.startswith("visual.")}
    patch = vis["visual.conv1.weight"].shape[-1]
    enc = ClipImageEncoder(vis, patch, prefix="visual.", precision=precision)
    n_px = patch * enc.grid
    out: Dict[str, torch.Tensor] = {}
    for i in range(0, len(p_images), batch_size):
        chunk = p_images[i:i + batch_size]
        x = torch.from_numpy(np.stack([_preprocess(p, n_px) for p in chunk])).to(device)
        emb = enc.encode_image(x).cpu()                                 # L2-normalised, fp32 (reference :72-76)
        for p, e in zip(chunk, emb):
            out[os.path.basename(p)] = e.clone()
        if fp is not None and ((i // batch_size) % max(1, (len(p_images) // batch_size) // 20) == 0 or i + batch_size >= len(p_images)):
            pkl.dump(out, open(fp, "wb"))                               # periodic checkpoint (reference :80-85)
    return out
