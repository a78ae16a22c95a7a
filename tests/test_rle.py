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
