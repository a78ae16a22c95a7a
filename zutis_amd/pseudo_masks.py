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


def _device_mask(engine: SelfMaskEngine, image: torch.Tensor, original_size: Optional[Tuple[int, int]],
                 bilateral_solver: bool) -> torch.Tensor:
    """image f32 [3,H,W] (normalised, on the GPU) -> uint8 {0,1} mask [H0,W0] on the GPU, all on the current stream."""
    out = engine.forward(image[None].contiguous(), inference=True)
    dt = out["dts"][0]                                                     # u8 [H,W] on device (selfmask.py:216-222)
    if bilateral_solver:                                                   # selfmask.py:226-234
        rgb = ops.denormalize_u8(image.contiguous())
        soft, _ = ops.bilateral_solve(rgb, dt.contiguous())
        dt = ops.threshold_f64_u8(soft, 0.5)                               # `> 0.5` on the device result
    if original_size is not None and tuple(original_size) != tuple(dt.shape):
        dt = ops.resize_nearest_u8(dt.contiguous(), int(original_size[0]), int(original_size[1]))   # index_dataset.py:215
    return dt


@torch.no_grad()
def pseudo_mask(engine: SelfMaskEngine, image: torch.Tensor, original_size: Optional[Tuple[int, int]] = None,
                bilateral_solver: bool = True) -> np.ndarray:
    """image f32 [3,H,W] (normalised, on the GPU) -> uint8 {0,1} mask [H0,W0] (original_size or H,W) on the host."""
    return _device_mask(engine, image, original_size, bilateral_solver).cpu().numpy()


@torch.no_grad()
def pseudo_masks_batch(engine: SelfMaskEngine, images: torch.Tensor, original_sizes: Optional[Sequence[Tuple[int, int]]] = None,
                       bilateral_solver: bool = True) -> List[torch.Tensor]:
    """B images of ONE size in one pass (the reference loops batch 1, index_dataset.py:189-204): images f32 [B,3,H,W] on the
    GPU -> list of B uint8 {0,1} masks on the GPU.  SelfMask runs batched; the bilateral solver's ~110 launches are shared by
    the whole batch (zh_bilateral_solve_batch, blockIdx.y = image) — for one image it is launch/latency-bound."""
    B = images.shape[0]
    out = engine.forward(images.contiguous(), inference=True)
    dts = out["dts"]                                                       # u8 [B,H,W]
    if bilateral_solver:
        rgb = torch.stack([ops.denormalize_u8(images[b].contiguous()) for b in range(B)])
        soft, _ = ops.bilateral_solve(rgb, dts.contiguous())
        dts = ops.threshold_f64_u8(soft, 0.5)
    masks = []
    for b in range(B):
        dt = dts[b]
        if original_sizes is not None and tuple(original_sizes[b]) != tuple(dt.shape):
            dt = ops.resize_nearest_u8(dt.contiguous(), int(original_sizes[b][0]), int(original_sizes[b][1]))
        masks.append(dt)
    return masks


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
def generate_pseudo_masks_batched(engine: SelfMaskEngine, images: Sequence[torch.Tensor], original_sizes: Sequence[Tuple[int, int]],
                                  out_paths: Sequence[str], bilateral_solver: bool = True, batch_size: int = 8) -> List[str]:
    """generate_pseudo_masks with images grouped by shape (consecutive images of one H x W, up to batch_size) so that SelfMask
    and the solver run batched.  Output files are identical to the batch-1 path (tests/test_bilateral_gpu.py)."""
    i, n = 0, len(images)
    while i < n:
        j = i + 1
        while j < n and j - i < batch_size and images[j].shape == images[i].shape:
            j += 1
        masks = pseudo_masks_batch(engine, torch.stack(list(images[i:j])), original_sizes[i:j], bilateral_solver)
        for m, path in zip(masks, out_paths[i:j]):
            save_rle_json(m.cpu().numpy(), path)
        i = j
    return list(out_paths)


@torch.no_grad()
def generate_pseudo_masks(engine: SelfMaskEngine, images: Sequence[torch.Tensor], original_sizes: Sequence[Tuple[int, int]],
                          out_paths: Sequence[str], bilateral_solver: bool = True, n_streams: int = 4) -> List[str]:
    """The loop of IndexDataset.generate_pseudo_masks (index_dataset.py:196-226) as a pipeline: image i runs SelfMask + solver +
    resize on HIP stream i % n_streams (one forked engine = one set of activation buffers per stream; the solver and the
    ViT of different images overlap — a single 512x683 image is launch/latency-bound: ~110 solver launches in 0.9 ms), the
    mask lands in pinned host memory by an async copy, and the host RLE-encodes / writes image i - n_streams meanwhile."""
    from collections import deque
    n_streams = max(1, min(n_streams, len(images)))
    if n_streams == 1:
        for img, size, path in zip(images, original_sizes, out_paths):
            save_rle_json(pseudo_mask(engine, img, size, bilateral_solver), path)
        return list(out_paths)
    dev = images[0].device
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
    engines = [engine] + [engine.fork() for _ in range(n_streams - 1)]
    pinned: List[Optional[torch.Tensor]] = [None] * n_streams
    pending = deque()

    def finish(item):
        ev, host, shape, path = item
        ev.synchronize()
        save_rle_json(host[: shape[0] * shape[1]].view(shape).numpy().copy(), path)

    for i, (img, size, path) in enumerate(zip(images, original_sizes, out_paths)):
        k = i % n_streams
        if len(pending) == n_streams:                                   # slot k's previous image: buffers free after this
            finish(pending.popleft())
        streams[k].wait_stream(torch.cuda.current_stream(dev))          # the image may have been produced on the caller's stream
        with torch.cuda.stream(streams[k]):
            dt = _device_mask(engines[k], img, size, bilateral_solver)
            n = dt.numel()
            if pinned[k] is None or pinned[k].numel() < n:
                pinned[k] = torch.empty((n,), dtype=torch.uint8, pin_memory=True)
            pinned[k][:n].copy_(dt.reshape(-1), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        pending.append((ev, pinned[k], tuple(dt.shape), path))
    while pending:
        finish(pending.popleft())
    return list(out_paths)


# ------------------------------------------------------------------------------------- the dataset method's own signature
def _image_size_hw(p_image: str) -> Tuple[int, int]:
    """(H, W) of the image file — `W, H = Image.open(p_image).size` (index_dataset.py:214)."""
    from PIL import Image
    with Image.open(p_image) as im:
        w, h = im.size
    return h, w


@torch.no_grad()
def dataset_generate_pseudo_masks(self, p_images: List[str], dir_dataset: str, n_workers: int = 4, bilateral_solver: bool = True, *,
                                  batch_size: int = 8, network=None, mask_dataset_cls=None, image_size_fn=None) -> None:
    """`IndexDataset.generate_pseudo_masks(self, p_images, dir_dataset, n_workers, bilateral_solver)` (datasets/index_dataset.py:177-226;
    the same method exists in datasets/imagenet.py and datasets/pass.py) over the batched device path.  Bind it in place of the
    reference's method — `IndexDataset.generate_pseudo_masks = zutis_amd.pseudo_masks.dataset_generate_pseudo_masks` — and the
    unchanged caller (`_get_pseudo_masks`, :263) runs SelfMask + bilateral solver + nearest resize on the GPU, batched by image
    shape, and writes the same JSON files to the same paths.

    What it takes from `self`, exactly as the reference's method does: `self.device` and `self._convert_p_image_to_p_pseudo_mask`.
    The loader is the dataset module's own `MaskDataset` (same resize / normalisation); the network is the `selfmask` of
    `utils.utils.get_network` (the overlay's drop-in SelfMask when it is installed) unless `network=` is given.  The keyword-only
    arguments are not in the reference's signature: `batch_size` (images of one shape solved together; 1 = the reference's loop),
    and the injection points the tests use."""
    import sys
    from torch.utils.data import DataLoader
    if network is None:
        from utils.utils import get_network                       # the reference's factory (utils/utils.py), as :184 calls it
        network = get_network(network_name="selfmask")
    network = network.to(self.device)
    network.eval()
    engine = network._get_engine() if hasattr(network, "_get_engine") else network
    if not isinstance(engine, SelfMaskEngine):
        raise TypeError("dataset_generate_pseudo_masks needs the MI355X drop-in SelfMask (networks/selfmask/selfmask.py of the overlay) "
                        "or a SelfMaskEngine: there is no torch / CPU fallback")
    if mask_dataset_cls is None:
        mask_dataset_cls = getattr(sys.modules[type(self).__module__], "MaskDataset")
    size_of = image_size_fn or _image_size_hw
    loader = DataLoader(dataset=mask_dataset_cls(p_images=p_images), batch_size=1, num_workers=n_workers, pin_memory=True)   # :187-188

    images: List[torch.Tensor] = []
    sizes: List[Tuple[int, int]] = []
    paths: List[str] = []

    def flush():
        if images:
            generate_pseudo_masks_batched(engine, images, sizes, paths, bilateral_solver=bilateral_solver, batch_size=batch_size)
            images.clear(); sizes.clear(); paths.clear()

    for dict_data in loader:                                        # :191-194: image 1 x 3 x H x W, p_image [str]
        image, p_image = dict_data["image"], dict_data["p_image"][0]
        if images and (tuple(image.shape[1:]) != tuple(images[0].shape) or len(images) >= batch_size):
            flush()                                                 # a group = consecutive images of one shape
        images.append(image[0].to(self.device, non_blocking=True).float())
        sizes.append(size_of(p_image))                              # :214 the file's own resolution
        paths.append(self._convert_p_image_to_p_pseudo_mask(p_image=p_image))   # :209
    flush()
    print(f"Pseudo-masks are saved in {dir_dataset}.")              # :226
