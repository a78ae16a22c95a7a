"""ORACLE — test infrastructure, NOT the product path (see oracle/zutis_ref.py header).

NumPy/SciPy restatement of the fast bilateral solver the reference applies to SelfMask pseudo-masks
(utils/bilateral_solver.py:21-195) and of the de-normalise-to-uint8 step in front of it
(utils/utils.py:261-273).  float64 throughout, fixed operation order.  Pinned against the real reference through
tests/golden/bilateral.npz (oracle/gen_golden.py; `cg(tol=)` shimmed to `rtol=` for SciPy >= 1.14).

The arithmetic is written in the *gather* form the HIP kernels use (per-vertex neighbour table instead of five CSR
matrices); the sums are taken in the same order as SciPy's CSR products, so grid quantities are bit-identical.
"""
from __future__ import annotations

import numpy as np

RGB_TO_YUV = np.array([[0.299, 0.587, 0.114], [-0.168736, -0.331264, 0.5], [0.5, -0.418688, -0.081312]])
YUV_OFFSET = np.array([0, 128.0, 128.0])


def denormalize_to_u8(x: np.ndarray, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)) -> np.ndarray:
    """utils/utils.py:261-273: fp32 x*std + mean, *255, clip, TRUNCATE to uint8, HWC.  x float32 [3,H,W]."""
    x = x.astype(np.float32) * np.asarray(std, np.float32)[:, None, None]
    x = x + np.asarray(mean, np.float32)[:, None, None]
    x = np.clip(x * np.float32(255), 0, 255)
    return x.astype(np.uint8).transpose(1, 2, 0)


def grid_coords(rgb: np.ndarray, sigma_spatial=16, sigma_luma=16, sigma_chroma=8) -> np.ndarray:
    """bilateral_solver.py:41-51: int coords [N,5] = (x/ss, y/ss, Y/sl, U/sc, V/sc), truncation toward zero."""
    h, w = rgb.shape[:2]
    yuv = np.tensordot(rgb, RGB_TO_YUV, ([2], [1])) + YUV_OFFSET.reshape(1, 1, -1)        # :21-22
    iy, ix = np.mgrid[:h, :w]
    c = np.dstack(((ix / sigma_spatial).astype(int), (iy / sigma_spatial).astype(int),
                   (yuv[..., 0] / sigma_luma).astype(int), (yuv[..., 1:] / sigma_chroma).astype(int)))
    return c.reshape(-1, 5)


class Grid:
    """bilateral_solver.py:58-85 in gather form: pix2v [N], neighbour table nbr [V,5,2] (-1 = absent), counts."""

    def __init__(self, rgb, sigma_spatial=16, sigma_luma=16, sigma_chroma=8):
        coords = grid_coords(rgb, sigma_spatial, sigma_luma, sigma_chroma)
        hv = 255 ** np.arange(5, dtype=np.int64)                       # :54 (exact: all hashes < 2^53)
        hashed = coords.astype(np.int64) @ hv
        self.uniq, first, self.pix2v = np.unique(hashed, return_index=True, return_inverse=True)
        self.npixels, self.nvertices = len(hashed), len(self.uniq)
        ucoords = coords[first]
        self.nbr = np.full((self.nvertices, 5, 2), -1, np.int64)
        for d in range(5):
            for s, off in enumerate((-1, 1)):
                nh = self.uniq + off * hv[d]
                loc = np.clip(np.searchsorted(self.uniq, nh), 0, self.nvertices - 1)       # get_valid_idx :29-37
                ok = self.uniq[loc] == nh
                self.nbr[ok, d, s] = loc[ok]
        self.coords, self.ucoords = coords, ucoords

    def splat(self, x):                                                # S.dot(x): sequential sum in pixel order
        out = np.zeros(self.nvertices)
        np.add.at(out, self.pix2v, x)
        return out

    def slice(self, y):
        return y[self.pix2v]

    def blur(self, x):                                                 # :94-100, sums in CSR column order (- then +)
        out = 2 * 5 * x
        for d in range(5):
            lo, hi = self.nbr[:, d, 0], self.nbr[:, d, 1]
            t = np.where(lo >= 0, x[np.maximum(lo, 0)], 0.0)
            t = np.where(hi >= 0, t + x[np.maximum(hi, 0)], t)
            out = out + t
        return out


def bistochastize(grid: Grid, maxiter=10):
    """:107-118"""
    m = grid.splat(np.ones(grid.npixels))
    n = np.ones(grid.nvertices)
    for _ in range(maxiter):
        n = np.sqrt(n * m / grid.blur(n))
    m = n * grid.blur(n)
    return n, m


def pcg(matvec, b, x0, minv, maxiter, rtol):
    """scipy.sparse.linalg.cg (1.15, atol=0) with a Jacobi preconditioner.  Returns (x, iterations run)."""
    if np.linalg.norm(b) == 0:                    # scipy: `if bnrm2 == 0: return postprocess(b), 0` — an empty target solves to zeros,
        return np.zeros_like(b), 0                # not to 0 / 0 (pinned by tests/golden/bilateral.npz "z_*": the reference on an all-zero target)
    x = x0.copy()
    r = b - matvec(x)
    atol = rtol * np.linalg.norm(b)
    p, rho_prev = None, None
    for it in range(maxiter):
        if np.linalg.norm(r) < atol:
            return x, it
        z = minv * r
        rho = np.dot(r, z)
        p = z.copy() if it == 0 else z + (rho / rho_prev) * p
        q = matvec(p)
        alpha = rho / np.dot(p, q)
        x += alpha * p
        r -= alpha * q
        rho_prev = rho
    return x, maxiter


def solve(grid: Grid, target: np.ndarray, confidence: np.ndarray, lam=256, a_diag_min=1e-5, cg_tol=1e-5, cg_maxiter=25):
    """:127-149 for one channel.  Returns (xhat [N], iterations, n, m)."""
    n, m = bistochastize(grid)
    w_splat = grid.splat(confidence)
    b = grid.splat(target * confidence)
    diag = lam * (m - n * 10.0 * n) + w_splat                          # A.diagonal(): blur's 2*dim centre weight

    def matvec(y):
        return lam * (m * y - n * grid.blur(n * y)) + w_splat * y
    minv = 1.0 / np.maximum(diag, a_diag_min)
    y0 = b / w_splat
    yhat, its = pcg(matvec, b, y0, minv, cg_maxiter, cg_tol)
    return grid.slice(yhat), its, n, m


def bilateral_solver_output(rgb: np.ndarray, target: np.ndarray, sigma_spatial=16, sigma_luma=16, sigma_chroma=8):
    """:152-195 -> (float64 soft [H,W], bool second-largest component [H,W])."""
    from scipy import ndimage
    h, w = target.shape
    grid = Grid(rgb, sigma_spatial, sigma_luma, sigma_chroma)
    t = target.reshape(-1).astype(np.double)
    c = np.ones(h * w) * 0.999
    soft = solve(grid, t, c)[0].reshape(h, w)
    return soft, postprocess(soft)


def postprocess(soft: np.ndarray) -> np.ndarray:
    """:185-193: fill holes, 4-connected labels, keep the SECOND largest label (index argsort[-2]); all-True fallback."""
    from scipy import ndimage
    binary = ndimage.binary_fill_holes(soft > 0.5)
    labeled, nr = ndimage.label(binary)
    nb = [np.sum(labeled == i) for i in range(nr + 1)]
    order = np.argsort(nb)
    try:
        return labeled == order[-2]
    except IndexError:
        return np.ones(soft.shape, dtype=bool)
