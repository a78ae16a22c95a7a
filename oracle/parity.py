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
