"""Size-independent properties of the host-side pieces of the path (hypothesis; CPU only, a few seconds): COCO-RLE round trips,
sharded top-k merging == global top-k, greedy mask NMS invariants, confusion-histogram additivity."""
import os
import sys

import numpy as np
from hypothesis import given, settings, strategies as st

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import zutis_ref as ref          # noqa: E402
from zutis_amd import rle                    # noqa: E402

FAST = settings(max_examples=40, deadline=None)


@FAST
@given(st.integers(1, 23), st.integers(1, 19), st.integers(0, 2 ** 32 - 1), st.sampled_from([0.0, 0.05, 0.5, 0.95, 1.0]))
def test_rle_round_trip_any_shape(h, w, seed, density):
    """networks/zutis.py:290,448 encode masks with pycocotools (column-major runs, zeros first); decode(encode(m)) == m, the C helper
    and the Python restatement agree, run lengths add up to H*W (empty and full masks included)."""
    m = (np.random.default_rng(seed).random((h, w)) < density).astype(np.uint8)
    r = rle.encode(m)
    assert r["size"] == [h, w]
    assert (rle.decode(r) == m).all()
    assert rle.encode_py(m)["counts"] == r["counts"]
    assert int(rle._counts(m).sum()) == h * w
    if m.any():
        ys, xs = np.nonzero(m)                       # torchvision.ops.masks_to_boxes convention: [xmin, ymin, xmax, ymax]
        assert rle.mask_to_box(m) == [float(xs.min()), float(ys.min()), float(xs.max()), float(ys.max())]


@FAST
@given(st.integers(1, 5), st.integers(1, 60), st.integers(1, 12), st.integers(1, 4), st.integers(0, 2 ** 32 - 1), st.booleans())
def test_sharded_topk_merge_equals_global_topk(C, N, k, shards, seed, ties):
    """Per-shard exact top-k + merge (zutis_amd/retrieval.py; oracle.merge_topk) == top-k over everything
    (datasets/index_dataset.py:163-167 with the restatement's tie order: score descending, index ascending)."""
    rng = np.random.default_rng(seed)
    D = 8
    text, img = rng.standard_normal((C, D)).astype(np.float32), rng.standard_normal((N, D)).astype(np.float32)
    if ties:
        img[rng.integers(0, N, size=max(1, N // 3))] = img[0]
    k = min(k, N)
    gi, gv = ref.retrieve_topk(text, img, k)
    sim = text @ img.T                                 # one product: BLAS blocking must not differ between the two sides
    topk = lambda s, kk: np.stack([np.lexsort((np.arange(s.shape[1]), -row))[:kk] for row in s])
    bounds = np.linspace(0, N, shards + 1).astype(int)
    ci, cv = [], []
    for a, b in zip(bounds[:-1], bounds[1:]):
        kk = min(k, b - a)
        if b > a:
            li = topk(sim[:, a:b], kk)
            lv = np.take_along_axis(sim[:, a:b], li, axis=1)
            li = li + a
        else:
            li, lv = np.zeros((C, 0), dtype=np.int64), np.zeros((C, 0), dtype=np.float32)
        pad = k - li.shape[1]
        ci.append(np.pad(li, ((0, 0), (0, pad)), constant_values=-1))
        cv.append(np.pad(lv, ((0, 0), (0, pad)), constant_values=-np.inf))
    # the merge breaks score ties by CANDIDATE column; shards are concatenated in index order, so that is index order too
    mi, mv = ref.merge_topk(np.concatenate(ci, 1), np.concatenate(cv, 1), k)
    assert np.array_equal(mv, gv) and np.array_equal(mi, gi)


@FAST
@given(st.integers(1, 9), st.integers(0, 2 ** 32 - 1), st.sampled_from(["hard", "linear", "gaussian"]))
def test_mask_nms_invariants(Q, seed, nms_type):
    """networks/zutis.py:211-299 greedy per-category NMS: kept indices are unique and valid, category 0 and empty masks never
    appear, each category's best-scoring non-empty mask is kept with its own score, and hard NMS is idempotent on its output."""
    rng = np.random.default_rng(seed)
    masks = rng.random((Q, 6, 7)) < rng.choice([0.0, 0.3, 0.7], size=(Q, 1, 1))
    scores = rng.random(Q)
    cats = rng.integers(0, 3, size=Q)
    out = ref.mask_nms(masks, scores, cats, nms_type=nms_type)
    idx = [m for _, m, _ in out]
    assert len(set(idx)) == len(idx) and all(0 <= m < Q for m in idx)
    assert all(c != 0 and masks[m].any() and cats[m] == c for c, m, _ in out)
    for c in set(int(v) for v in cats) - {0}:
        members = np.nonzero(cats == c)[0]
        best = members[np.argmax(scores[members])]
        if masks[best].any():
            assert (c, int(best), float(scores[best])) in out
    if nms_type == "hard" and out:
        keep = np.array(idx)
        again = ref.mask_nms(masks[keep], np.array([s for _, _, s in out]), cats[keep], nms_type="hard")
        assert sorted(m for _, m, _ in again) == list(range(len(keep)))


@FAST
@given(st.integers(2, 9), st.integers(1, 40), st.integers(1, 40), st.integers(0, 2 ** 32 - 1))
def test_confusion_hist_is_additive_and_scores_are_bounded(n, a, b, seed):
    """utils/running_score.py:11-16,22-49: the histogram of a concatenation is the sum of the histograms (what lets ranks
    all-reduce it); out-of-range ground-truth labels are ignored; every score lies in [0, 1]."""
    rng = np.random.default_rng(seed)
    t1, p1 = rng.integers(-1, n + 1, size=a), rng.integers(0, n, size=a)
    t2, p2 = rng.integers(-1, n + 1, size=b), rng.integers(0, n, size=b)
    h1, h2 = ref.confusion_hist(t1, p1, n), ref.confusion_hist(t2, p2, n)
    h = ref.confusion_hist(np.concatenate([t1, t2]), np.concatenate([p1, p2]), n)
    assert np.array_equal(h, h1 + h2)
    assert h.sum() == ((t1 >= 0) & (t1 < n)).sum() + ((t2 >= 0) & (t2 < n)).sum()
    if h.sum() > 0:
        sc, _ = ref.scores_from_hist(h)
        assert all(0.0 <= v <= 1.0 + 1e-12 for v in sc.values() if not np.isnan(v))


def test_label_mismatch_explanation_helper():
    """oracle/parity.py: labels computed from logits perturbed by <= err differ from the reference's only where the reference's
    top-2 margin is <= 2 err (zero unexplained); a label flipped on a confident pixel is flagged."""
    import numpy as np
    from oracle import resample as R
    from oracle.parity import unexplained_label_mismatches
    rng = np.random.default_rng(0)
    lo = (rng.standard_normal((2, 9, 10, 14)) * 0.05).astype(np.float32)     # near-ties everywhere: many flips
    err = 0.02
    lo2 = (lo + rng.uniform(-err, err, lo.shape)).astype(np.float32)
    ref, got = R.bilinear_argmax_nchw(lo, 80, 112), R.bilinear_argmax_nchw(lo2, 80, 112)
    e = float(np.abs(lo2 - lo).max())
    n_mis, n_bad, worst = unexplained_label_mismatches(got, ref, lo, e, (80, 112))
    assert n_mis > 100 and n_bad == 0 and worst <= 2 * e + 2e-6
    full = R.bilinear_nchw(lo[:1], 80, 112)[0]
    srt = np.sort(full, axis=0)
    y, x = np.unravel_index(np.argmax(srt[-1] - srt[0]), srt[-1].shape)      # the pixel with the widest spread: flip it to its worst class
    bad = ref.copy()
    bad[0, y, x] = int(np.argmin(full[:, y, x]))
    assert unexplained_label_mismatches(bad, ref, lo, 1e-4, (80, 112))[1] == 1


def test_long_sequence_key_split_model():
    """engine_base.long_sequence_key_split (round 6): the split is 1 .. 8, never leaves a workgroup without keys, takes 1 when the grid
    already fills whole rounds of the chip, and splits when the launch is one under-filled round (SelfMask at T = 5505: one image = 264
    workgroups on 256 CUs; four images = 1.375 rounds of 768 slots) — the picks measured in profiles/r06_attn_long_split.txt."""
    from zutis_amd.engine_base import long_sequence_key_split as f
    T, H, dh = 5505, 6, 64
    nqb = -(-T // 128)
    picks = {B: f(B * H * nqb, -(-T // 32), dh, True, B * T * H * dh) for B in (1, 2, 4, 8)}
    assert picks == {1: 5, 2: 4, 4: 2, 8: 1}, picks
    for B in (1, 2, 3, 4, 5, 8, 16):
        for x3 in (True, False):
            kt = -(-T // (32 if x3 else 64))
            S = f(B * H * nqb, kt, dh, x3, B * T * H * dh)
            assert 1 <= S <= 8 and (S == 1 or (S - 1) * -(-kt // S) < kt)
    assert f(768 * 4, 173, 64, True, 10 ** 9) == 1              # whole rounds already: nothing to gain, the merge only costs
    assert f(96, 300, 96, True, 10 ** 6) > 1                    # dh = 96: two workgroups per CU, 96 items leave most CUs idle


def test_natural_images_are_deterministic_and_normalised():
    """benchlib.objects.natural_images (the pseudo-label bench / tests workload): the same tensor every time, ImageNet-normalised u8 values."""
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    a = bench.natural_images(2, 40, 56, "cpu", seed=7)
    b = bench.natural_images(2, 40, 56, "cpu", seed=7)
    assert a.shape == (2, 3, 40, 56) and a.dtype == torch.float32 and torch.equal(a, b) and not torch.equal(a[0], a[1])
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    u8 = (a * std + mean) * 255.0
    assert float((u8 - u8.round()).abs().max()) < 1e-3 and 0 <= float(u8.min()) and float(u8.max()) <= 255.001
