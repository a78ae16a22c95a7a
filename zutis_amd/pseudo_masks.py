"""Pseudo-label driver: SelfMask + bilateral solver + nearest resize + COCO-RLE JSON, with the mask
staying on the GPU between the stages.

Mirrors IndexDataset.generate_pseudo_masks (datasets/index_dataset.py:177-226): the reference runs batch 1, pulls the
mask to the host (`.cpu()`), converts the image tensor to PIL on the host, solves in NumPy/SciPy, re-uploads nothing and
resizes with F.interpolate on CPU tensors.  Here: SelfMaskEngine.forward(inference=True) -> device u8 mask ->
zh_denormalize_u8 + zh_bilateral_solve (device) -> `> 0.5` -> zh_resize_nearest_u8 (device) -> one D2H -> RLE.
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops, rle
from .engine import SelfMaskEngine


@torch.no_grad()
def pseudo_mask(engine: SelfMaskEngine, image: torch.Tensor, original_size: Optional[Tuple[int, int]] = None,
                bilateral_solver: bool = True) -> np.ndarray:
    """image f32 [3,H,W] (normalised, on the GPU) -> uint8 {0,1} mask [H0,W0] (original_size or H,W) on the host."""
    out = engine.forward(image[None].contiguous(), inference=True)
    dt = out["dts"][0]                                                     # u8 [H,W] on device (selfmask.py:216-222)
    if bilateral_solver:                                                   # selfmask.py:226-234
        rgb = ops.denormalize_u8(image.contiguous())
        soft, _ = ops.bilateral_solve(rgb, dt.contiguous())
        dt = (soft > 0.5).to(torch.uint8)                                  # comparison on the device result
    if original_size is not None and tuple(original_size) != tuple(dt.shape):
        dt = ops.resize_nearest_u8(dt.contiguous(), int(original_size[0]), int(original_size[1]))   # index_dataset.py:215
    return dt.cpu().numpy()


def save_rle_json(mask: np.ndarray, path: str) -> Dict:
    """index_dataset.py:219-224: RLE-encode (Fortran order), dump as JSON (counts as str, like ujson reject_bytes=False),
    read back and assert the round trip."""
    r = rle.encode(mask)
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with open(path, "w") as f:
        json.dump({"size": r["size"], "counts": r["counts"].decode("ascii")}, f)
    back = json.load(open(path))
    assert (rle.decode(back) == mask).sum() == mask.size
    return r


@torch.no_grad()
def generate_pseudo_masks(engine: SelfMaskEngine, images: Sequence[torch.Tensor], original_sizes: Sequence[Tuple[int, int]],
                          out_paths: Sequence[str], bilateral_solver: bool = True) -> List[str]:
    for img, size, path in zip(images, original_sizes, out_paths):
        save_rle_json(pseudo_mask(engine, img, size, bilateral_solver), path)
    return list(out_paths)
