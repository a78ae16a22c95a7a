import numpy as np

from zutis_amd import rle, detgen


def test_rle_round_trip_and_format():
    for seed, shape in enumerate([(7, 5), (80, 112), (1, 1), (33, 64)]):
        m = detgen.det_normal("rle", shape, seed=seed) > 0.3
        r = rle.encode(m)
        assert r["size"] == list(shape) and isinstance(r["counts"], bytes)
        assert all(48 <= c < 48 + 64 for c in r["counts"])
        assert np.array_equal(rle.decode(r).astype(bool), m)
    z = np.zeros((4, 6), bool)
    assert np.array_equal(rle.decode(rle.encode(z)), z)
    o = np.ones((4, 6), bool)
    assert rle._counts(o).tolist() == [0, 24]                       # leading zero-run of length 0
    assert np.array_equal(rle.decode(rle.encode(o)).astype(bool), o)
    # long runs and negative deltas exercise the sign-extension branch
    big = np.zeros((300, 300), bool)
    big[5:290, 7] = True
    big[0:3, 200] = True
    assert np.array_equal(rle.decode(rle.encode(big)).astype(bool), big)


def test_known_counts_string():
    # column-major runs of [[0,1],[1,1]] are 1 zero, 3 ones -> counts [1,3] -> chars '1','3'
    assert rle.encode(np.array([[0, 1], [1, 1]], bool))["counts"] == b"13"


def test_mask_to_box():
    m = np.zeros((10, 12), bool)
    m[2:5, 3:9] = True
    assert rle.mask_to_box(m) == [3.0, 2.0, 8.0, 4.0]


def test_c_helper_matches_python_restatement():
    for seed, shape in enumerate([(7, 5), (80, 112), (1, 1), (33, 64), (336, 336)]):
        m = detgen.det_normal("rlec", shape, seed=seed) > 0.8
        assert rle.encode(m)["counts"] == rle.encode_py(m)["counts"]
    for m in (np.zeros((4, 6), bool), np.ones((4, 6), bool)):
        assert rle.encode(m)["counts"] == rle.encode_py(m)["counts"]
    big = np.zeros((300, 300), bool); big[5:290, 7] = True; big[0:3, 200] = True
    assert rle.encode(big)["counts"] == rle.encode_py(big)["counts"]


def _mask_from_runs(size, runs):
    h, w = size
    flat = np.zeros(h * w, np.uint8)
    pos, val = 0, 0
    for r in runs:
        flat[pos:pos + r] = val
        pos += r
        val ^= 1
    assert pos == h * w
    return flat.reshape((h, w), order="F")


def test_rle_hand_derived_format_vectors(golden_dir):
    """RLE byte strings pinned to vectors derived BY HAND from the published COCO format (pycocotools maskApi.c rleToString;
    the package is absent), tests/golden/rle_vectors.json.  Worked examples (c = x & 31; x >>= 5; more = bit4(c) ? x != -1 :
    x != 0; char = (c | 32*more) + 48; counts i > 2 stored minus counts[i-2]):
      9            -> c=9, x=0, stop                      -> '9'
      40           -> c=8, x=1, more -> 8|32=40 -> 'X'; then c=1 -> '1'            -> "X1"
      1 - 2 = -1   -> c=31, x=-1, bit4 set and x == -1 -> stop -> 31+48 -> 'O'
      1000         -> c=8,x=31 -> 'X'; c=31,x=0, bit4 set, 0 != -1 -> more -> 63+48 -> 'o'; c=0 -> '0'   -> "Xo0"
      3 - 1000     -> -997: c=27,x=-32 -> 27|32 -> 'k'; c=0,x=-1, more (x != 0) -> 'P'; c=31,x=-1 stop -> 'O' -> "kPO"
      2000 - 5     -> 1995: c=11,x=62 -> '['; c=30,x=1, bit4 set, more -> 'n'; c=1 -> '1'   -> "[n1"
      31           -> c=31,x=0, bit4 set and 0 != -1 -> more -> 'o'; then '0' (a positive value whose top group has bit 4 set
                      needs one more all-zero group so that the decoder does not sign-extend it)
      16           -> c=16 -> same guard -> "`0"
    """
    import json
    vecs = json.load(open(f"{golden_dir}/rle_vectors.json"))["vectors"]
    assert len(vecs) >= 7
    for v in vecs:
        m = _mask_from_runs(v["size"], v["runs_colmajor"])
        want = v["counts"].encode("ascii")
        assert rle._counts(m).tolist() == v["runs_colmajor"], v["name"]
        assert rle.encode_py(m)["counts"] == want, v["name"]                 # NumPy/Python restatement
        assert rle.encode(m)["counts"] == want, v["name"]                    # C helper in libzutis_hip.so
        assert rle.encode(m)["size"] == v["size"]
        assert np.array_equal(rle.decode({"size": v["size"], "counts": want}), m), v["name"]
        assert rle._from_string(want) == v["runs_colmajor"], v["name"]


def test_batched_rle_from_transitions_host_entry():
    """zh_rle_from_transitions_host (all kept masks' strings in one C call, from the device's transition positions) == the per-mask
    path == the mask encoder: random, empty, full, a rectangle, a mask with pixel 0 set; a mask over the position capacity is
    reported as None (the caller re-encodes it)."""
    import numpy as np
    from zutis_amd import rle
    rng = np.random.default_rng(0)
    H, W = 37, 53
    masks = [(rng.random((H, W)) > 0.5).astype(np.uint8), np.zeros((H, W), np.uint8), np.ones((H, W), np.uint8)]
    m = np.zeros((H, W), np.uint8); m[5:20, 10:30] = 1; masks.append(m)
    m = np.zeros((H, W), np.uint8); m[0, 0] = 1; masks.append(m)
    pos, nr = [], []
    for m in masks:
        f = m.reshape(-1, order="F")
        t = np.flatnonzero(f[1:] != f[:-1]) + 1
        pos.append(t); nr.append((len(t), int(f[0])))
    keep = max(1, max(len(t) for t in pos))
    P = np.zeros((len(masks), keep), np.int32)
    for i, t in enumerate(pos):
        P[i, :len(t)] = t
    NR = np.array(nr, np.int32)
    out = rle.rles_from_transitions(P, NR, H, W)
    for i, m in enumerate(masks):
        assert out[i] == rle.encode(m) == rle.rle_from_transitions(pos[i], nr[i][1], H, W), i
        assert (rle.decode(out[i]) == m).all()
    short = rle.rles_from_transitions(P[:, :3], NR, H, W)
    assert short[0] is None and short[1] == rle.encode(masks[1])
    # the packed list of zh_mask_runs_kept: every mask's min(transitions, max_runs) entries back to back
    for max_runs in (keep, 3):
        flat = np.concatenate([t[:max_runs] for t in pos]).astype(np.int32)
        got = rle.rles_from_transitions(flat, NR, H, W, packed_max_runs=max_runs)
        for i, m in enumerate(masks):
            assert got[i] == (rle.encode(m) if len(pos[i]) <= max_runs else None), (max_runs, i)
