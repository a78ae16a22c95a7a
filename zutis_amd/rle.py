"""COCO run-length encoding and mask boxes without pycocotools / torchvision.

The reference calls pycocotools.mask.encode(np.asfortranarray(m)) (networks/zutis.py:290,448) and
torchvision.ops.masks_to_boxes (zutis.py:294,452).  Neither package is in this image, so their published
algorithms are restated here (pycocotools 2.0 maskApi.c: rleEncode + rleToString; torchvision.ops.boxes.masks_to_boxes).
The RLE byte string is pinned by hand-derived vectors of the published format (tests/golden/rle_vectors.json, worked out in
tests/test_rle.py: multi-character values, negative deltas, the sign-guard group) plus decode(encode(m)) == m; pycocotools
itself is not available to compare with.  When pycocotools IS importable, networks.zutis uses it instead.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np


def _counts(mask: np.ndarray) -> np.ndarray:
    """Run lengths of the column-major flattened mask, starting with the run of zeros (may be 0)."""
    flat = np.asarray(mask, dtype=np.uint8).reshape(-1, order="F")
    if flat.size == 0:
        return np.zeros((0,), np.int64)
    change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    bounds = np.concatenate(([0], change, [flat.size]))
    runs = np.diff(bounds)
    if flat[0] != 0:
        runs = np.concatenate(([0], runs))
    return runs.astype(np.int64)


def _to_string(cnts: np.ndarray) -> bytes:
    out = bytearray()
    for i, c in enumerate(cnts.tolist()):
        x = int(c)
        if i > 2:
            x -= int(cnts[i - 2])
        more = True
        while more:
            ch = x & 0x1F
            x >>= 5
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            out.append(ch + 48)
    return bytes(out)


def _from_string(s: bytes) -> List[int]:
    cnts: List[int] = []
    p = 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = s[p] - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(cnts) > 2:
            x += cnts[-2]
        cnts.append(x)
    return cnts


def encode_py(mask: np.ndarray) -> Dict:
    """Pure NumPy/Python form (kept as the readable restatement and as the checker of the C helper)."""
    assert mask.ndim == 2
    h, w = mask.shape
    return {"size": [int(h), int(w)], "counts": _to_string(_counts(mask))}


def encode(mask: np.ndarray) -> Dict:
    """mask [H,W] {0,1}/bool -> {"size": [H, W], "counts": bytes} (pycocotools.mask.encode for one mask).
    Uses the C helper zh_rle_encode_host from libzutis_hip.so (host code, no GPU needed)."""
    import ctypes as C
    from . import _lib
    assert mask.ndim == 2
    h, w = mask.shape
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    cap = 8 * (m.size + 2) // 2 + 16
    cap = min(cap, 6 * (h * w + 2))
    buf = C.create_string_buffer(cap)
    n = _lib.load().zh_rle_encode_host(m.ctypes.data, h, w, C.addressof(buf), cap)
    if n < 0:
        raise RuntimeError("zh_rle_encode_host: buffer too small")
    return {"size": [int(h), int(w)], "counts": buf.raw[:n]}


def decode(rle: Dict) -> np.ndarray:
    h, w = rle["size"]
    cnts = _from_string(rle["counts"] if isinstance(rle["counts"], (bytes, bytearray)) else rle["counts"].encode("ascii"))
    flat = np.zeros(h * w, np.uint8)
    pos, val = 0, 0
    for c in cnts:
        if val:
            flat[pos:pos + c] = 1
        pos += c
        val ^= 1
    return flat.reshape((h, w), order="F")


def mask_to_box(mask: np.ndarray) -> List[float]:
    """torchvision.ops.masks_to_boxes for one mask: [xmin, ymin, xmax, ymax] of the non-zero pixels (float32 values)."""
    rows = np.flatnonzero(mask.any(axis=1))
    cols = np.flatnonzero(mask.any(axis=0))
    return [float(cols[0]), float(rows[0]), float(cols[-1]), float(rows[-1])]


def rle_from_transitions(positions: np.ndarray, first_value: int, h: int, w: int) -> Dict:
    """COCO RLE dict from the column-major transition positions produced by zh_mask_runs (device):
    counts = diff([0, positions..., h*w]) with a leading empty zero-run when pixel 0 is set."""
    import ctypes as C
    from . import _lib
    edges = np.concatenate(([0], positions.astype(np.int64), [h * w]))
    counts = np.diff(edges)
    if first_value:
        counts = np.concatenate(([0], counts))
    counts = np.ascontiguousarray(counts, dtype=np.int64)
    cap = 8 * counts.size + 16
    buf = C.create_string_buffer(cap)
    n = _lib.load().zh_rle_counts_to_string_host(counts.ctypes.data, counts.size, C.addressof(buf), cap)
    assert n >= 0
    return {"size": [int(h), int(w)], "counts": buf.raw[:n]}


def rles_from_transitions(positions: np.ndarray, nruns: np.ndarray, h: int, w: int, packed_max_runs: int = 0):
    """The COCO RLE dicts of all n masks from zh_mask_runs' host copies in ONE C call (zh_rle_from_transitions_host): positions int32
    [n, keep], nruns int32 [n, 2] = (transitions, value of pixel 0).  Entry i is None when mask i has more transitions than `keep`
    (the caller re-encodes it from the mask itself).  packed_max_runs > 0: positions is zh_mask_runs_kept's packed list (1-D: mask i's
    min(transitions, packed_max_runs) entries follow mask i - 1's)."""
    import ctypes as C
    from . import _lib
    pos = positions if (positions.dtype == np.int32 and positions.flags.c_contiguous) else np.ascontiguousarray(positions, dtype=np.int32)
    nr = nruns if (nruns.dtype == np.int32 and nruns.flags.c_contiguous) else np.ascontiguousarray(nruns, dtype=np.int32)
    n = nr.shape[0]
    keep = int(packed_max_runs) if packed_max_runs else pos.shape[1]
    nt = nr[:, 0].tolist()
    cap = 8 * (sum(min(t, keep) for t in nt) + 3 * n) + 16
    buf = C.create_string_buffer(cap)
    off = np.empty(n + 1, dtype=np.int64)
    total = _lib.load(raw=True).zh_rle_from_transitions_host(pos.ctypes.data, keep, 1 if packed_max_runs else 0, nr.ctypes.data, n, h * w,
                                                             C.addressof(buf), cap, off.ctypes.data)
    assert total >= 0
    raw = buf.raw
    size = [int(h), int(w)]
    o = off.tolist()
    return [({"size": size, "counts": raw[o[i]:o[i + 1]]} if nt[i] <= keep else None) for i in range(n)]
