"""ORACLE — test infrastructure, NOT the product path (see oracle/zutis_ref.py header).

NumPy float32 restatements of the ATen CPU resampling kernels the reference reaches through
F.interpolate (networks/zutis.py:368,422,492; networks/clip_arch.py:366-371).

Bit-exactness notes (probed against torch 2.10 CPU in the authoring container, pinned by
tests/test_oracle_resample.py):
  * ATen's kernels are compiled with FMA contraction.  Bilinear (upsample_generic_Nd_kernel_impl):
        src   = max(fma(scale, dst + 0.5, -0.5), 0)            scale = float(in)/float(out)   (size= form)
        i0    = min(int(src), in-1);  i1 = min(i0+1, in-1);  l1 = clamp(src - i0, 0, 1);  l0 = 1 - l1
        row_y = fma(v[y][x0], lx0, v[y][x1] * lx1)
        out   = fma(row_y0, ly0, row_y1 * ly1)
    reproduces torch bit-for-bit on every shape tried.
  * fma() is emulated as float32(float64(a)*float64(b) + float64(c)); the product is exact in float64,
    so the only deviation from a true fma is a double rounding with probability ~2^-29 per operation.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def fma(a, b, c):
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(np.float32)


def linear_index_weights(in_size: int, out_size: int, scale=None):
    """ATen compute_source_index_and_lambda (align_corners=False).  `scale` = 1/scale_factor when the caller
    passed scale_factor, else in/out."""
    if out_size == in_size:
        i = np.arange(out_size, dtype=np.int64)
        return i, i, np.ones(out_size, np.float32), np.zeros(out_size, np.float32)
    sc = f32(in_size) / f32(out_size) if scale is None else f32(scale)
    d = np.arange(out_size, dtype=np.float32)
    src = np.maximum(fma(sc, d + f32(0.5), f32(-0.5)), f32(0))
    i0 = np.minimum(src.astype(np.int64), in_size - 1)
    l1 = np.clip(src - i0.astype(np.float32), f32(0), f32(1)).astype(np.float32)
    i1 = np.minimum(i0 + 1, in_size - 1)
    return i0, i1, (f32(1) - l1).astype(np.float32), l1


def bilinear_nchw(x: np.ndarray, H: int, W: int) -> np.ndarray:
    """F.interpolate(x[B,C,h,w], size=(H,W), mode='bilinear') — networks/zutis.py:368,422."""
    x = np.ascontiguousarray(x, np.float32)
    y0, y1, ly0, ly1 = linear_index_weights(x.shape[2], H)
    x0, x1, lx0, lx1 = linear_index_weights(x.shape[3], W)
    top, bot = x[:, :, y0], x[:, :, y1]
    r0 = fma(top[..., x0], lx0, top[..., x1] * lx1)
    r1 = fma(bot[..., x0], lx0, bot[..., x1] * lx1)
    return fma(r0, ly0[:, None], r1 * ly1[:, None])


def bilinear_argmax_nchw(x: np.ndarray, H: int, W: int) -> np.ndarray:
    """argmax over C of bilinear_nchw (first index on ties, as torch.argmax on CPU) -> int64 [B,H,W].
    networks/zutis.py:366-372."""
    out = np.empty((x.shape[0], H, W), np.int64)
    for b in range(x.shape[0]):                       # one image at a time: bounded memory
        out[b] = np.argmax(bilinear_nchw(x[b:b + 1], H, W)[0], axis=0)
    return out


def bilinear_up2_cl(x: np.ndarray) -> np.ndarray:
    """F.interpolate(scale_factor=2, mode='bilinear') on channels-last [B,h,w,C] -> [B,2h,2w,C].
    networks/zutis.py:491-495 (the reference permutes to NCHW and back; values are identical)."""
    x = np.ascontiguousarray(x, np.float32)
    B, h, w, C = x.shape
    y0, y1, ly0, ly1 = linear_index_weights(h, 2 * h, scale=0.5)
    x0, x1, lx0, lx1 = linear_index_weights(w, 2 * w, scale=0.5)
    top, bot = x[:, y0], x[:, y1]
    r0 = fma(top[:, :, x0], lx0[:, None], top[:, :, x1] * lx1[:, None])
    r1 = fma(bot[:, :, x0], lx0[:, None], bot[:, :, x1] * lx1[:, None])
    return fma(r0, ly0[None, :, None, None], r1 * ly1[None, :, None, None])


# ----------------------------------------------------------------------------- bicubic
_A = f32(-0.75)


def _cc1(x):  # cubic_convolution1: ((A+2)x - (A+3)) x^2 + 1
    return (((_A + f32(2)) * x - (_A + f32(3))) * x * x + f32(1)).astype(np.float32)


def _cc2(x):  # cubic_convolution2: ((A x - 5A) x + 8A) x - 4A
    return (((_A * x - f32(5) * _A) * x + f32(8) * _A) * x - f32(4) * _A).astype(np.float32)


def cubic_index_weights(in_size: int, out_size: int, scale):
    """ATen HelperInterpCubic::compute_indices_weights, align_corners=False: src = scale*(dst+.5)-.5 (no clamp),
    taps floor(src)-1..+2 clamped to [0,in-1], Keys A=-0.75."""
    sc = f32(scale)
    d = np.arange(out_size, dtype=np.float32)
    src = fma(sc, d + f32(0.5), f32(-0.5))
    fl = np.floor(src)
    t = (src - fl).astype(np.float32)
    i = fl.astype(np.int64)
    idx = np.stack([np.clip(i + k, 0, in_size - 1) for k in (-1, 0, 1, 2)], axis=0)
    wts = np.stack([_cc2(t + f32(1)), _cc1(t), _cc1(f32(1) - t), _cc2(f32(2) - t)], axis=0)
    return idx, wts


def bicubic_cl(x: np.ndarray, H: int, W: int, scale_factor_h=None, scale_factor_w=None) -> np.ndarray:
    """upsample_bicubic2d on a channels-last [h,w,C] grid -> [H,W,C].  With scale_factor given the coordinate
    scale is float32(1.0/double(scale_factor)) (ATen area_pixel_compute_scale), else in/out.
    networks/clip_arch.py:366-371 passes scale_factor=((h+.1)/g,(w+.1)/g) so scale = g/(h+.1) — NOT g/h
    (SURVEY.md Appendix A); networks/selfmask/vision_transformer.py:392-397 passes size= (scale = in/out)."""
    x = np.ascontiguousarray(x, np.float32)
    sh = f32(1.0 / float(scale_factor_h)) if scale_factor_h is not None else f32(x.shape[0]) / f32(H)
    sw = f32(1.0 / float(scale_factor_w)) if scale_factor_w is not None else f32(x.shape[1]) / f32(W)
    iy, wy = cubic_index_weights(x.shape[0], H, sh)
    ix, wx = cubic_index_weights(x.shape[1], W, sw)
    out = np.zeros((H, W, x.shape[2]), np.float32)
    for a in range(4):
        rows = x[iy[a]]                                            # [H,w,C]
        acc = rows[:, ix[0]] * wx[0][None, :, None]
        for b in range(1, 4):
            acc = fma(rows[:, ix[b]], wx[b][None, :, None], acc)
        out = acc * wy[a][:, None, None] if a == 0 else fma(acc, wy[a][:, None, None], out)
    return out
