"""ORACLE-side helper (test infrastructure: tests/ and bench.py's parity leg only) — label parity stated honestly (north_star: "bit-exact for mask indexing/argmax").

The fused upsample+argmax kernel is bit-exact GIVEN equal low-res logits (tests/test_kernels_gpu.py); what can differ between
the HIP path and the reference is the logits themselves, by at most `err` = max |lo - lo_ref|.  Bilinear interpolation is a
convex combination, so the full-resolution logits differ by at most `err` too, and a pixel can legitimately carry another
label than the reference's ONLY if the reference's own full-resolution logits separate the two labels by at most 2 * err:
    L_hip[g] >= L_hip[r]  =>  L_ref[g] + err >= L_ref[r] - err  =>  L_ref[r] - L_ref[g] <= 2 err.
`unexplained_label_mismatches` counts the pixels that break this; the tests assert it is ZERO instead of asserting an
agreement fraction.
"""
import numpy as np

from . import resample as R

SLACK = 2e-6     # fp32 rounding of the interpolation itself (four products and three sums per value), on |logit| <= ~1 values


def unexplained_label_mismatches(labels, labels_ref, logits_lo_ref, err, size):
    """labels / labels_ref int [B,H,W]; logits_lo_ref f32 [B,n,h,w] (the reference's low-res logits); err = max abs low-res logit
    error of the path under test.  Returns (mismatching pixels, unexplained ones, largest reference margin among mismatches)."""
    labels, labels_ref = np.asarray(labels).astype(np.int64), np.asarray(labels_ref).astype(np.int64)
    n_mis = n_bad = 0
    worst = 0.0
    for b in range(labels.shape[0]):
        ys, xs = np.nonzero(labels[b] != labels_ref[b])
        if len(ys) == 0:
            continue
        full = R.bilinear_nchw(np.ascontiguousarray(logits_lo_ref[b:b + 1]), int(size[0]), int(size[1]))[0]   # [n,H,W], ATen-exact
        margin = full[labels_ref[b, ys, xs], ys, xs] - full[labels[b, ys, xs], ys, xs]
        n_mis += len(ys)
        n_bad += int((margin > 2.0 * err + SLACK).sum())
        worst = max(worst, float(margin.max()))
    return n_mis, n_bad, worst


def nearest_resize(mask, H0, W0):
    """F.interpolate(mode="nearest") of a 2-D array (datasets/index_dataset.py:215): source index = floor(dst * float32(in / out))."""
    H, W = mask.shape
    return mask[(np.arange(H0) * np.float32(H / H0)).astype(int)][:, (np.arange(W0) * np.float32(W / W0)).astype(int)]


def pseudo_mask_chain(P, x, out_size, bilateral_solver=True):
    """The oracle's pseudo-label chain for ONE normalised image x [1,3,H,W] (torch, CPU): SelfMask inference (selfmask.py:204-224) ->
    bilateral solver on the denormalised image (selfmask.py:226-234, utils/bilateral_solver.py) -> > 0.5 -> nearest resize to the file's
    resolution (datasets/index_dataset.py:214-215).  Returns {"mask" bool [H0,W0], "mask_native" bool [H,W] (before the resize), "selfmask" bool [H,W], "query", "objectness_margin"}."""
    import torch
    from . import bilateral_ref as B
    from . import selfmask_ref as S
    with torch.no_grad():
        out = S.selfmask_forward(P, x)
        dts, idx, _ = S.selfmask_inference(P, x, out=out)
    ol = np.sort(out["objectness_logits"][0].numpy())
    m = dts[0]
    sm = m.astype(bool)
    if bilateral_solver:
        soft, _ = B.bilateral_solver_output(B.denormalize_to_u8(x[0].numpy()), m)
        m = soft > 0.5
    m = np.asarray(m).astype(bool)
    return {"mask": nearest_resize(m, int(out_size[0]), int(out_size[1])), "mask_native": m, "selfmask": sm, "query": int(idx[0]),
            "objectness_margin": float(ol[-1] - ol[-2])}


def contour_mismatches(got, ref, native=None):
    """Binary masks [H,W]: (differing pixels, those NOT within one pixel of the reference's own contour).  A pixel is on the contour
    when its 3 x 3 neighbourhood in the reference holds both values — where a soft output within rounding of the 0.5 threshold may fall
    either way.  `native`: the reference mask BEFORE a nearest resize to got's shape — the contour is then taken there and carried
    through the same resize (one source pixel becomes a block of pixels; all of the block is 'on the contour')."""
    got, ref = np.asarray(got).astype(bool), np.asarray(ref).astype(bool)
    src = ref if native is None else np.asarray(native).astype(bool)
    p = np.pad(src, 1, mode="edge")
    H, W = src.shape
    nb = np.stack([p[dy:dy + H, dx:dx + W] for dy in range(3) for dx in range(3)])
    contour = nb.any(axis=0) != nb.all(axis=0)
    if native is not None:
        contour = nearest_resize(contour, ref.shape[0], ref.shape[1])
    diff = got != ref
    return int(diff.sum()), int((diff & ~contour).sum())
