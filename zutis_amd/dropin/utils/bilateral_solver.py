"""Drop-in replacement for the reference's utils/bilateral_solver.py on MI355X.

`bilateral_solver_output(img, target, sigma_spatial=16, sigma_luma=16, sigma_chroma=8)` keeps the reference signature and
return types (utils/bilateral_solver.py:152-195): (float64 [H,W] soft output, bool [H,W] second-largest component).
The grid build, bistochastisation and PCG solve run in float64 HIP kernels (zutis_amd/csrc/bilateral.hip); the
hole-filling / connected-component post-processing — whose result SelfMask discards (selfmask.py:230-231) — stays on
the host with scipy.ndimage exactly as in the reference.
"""
import numpy as np
import torch
from scipy import ndimage

from zutis_amd import ops as _ops


def _postprocess(output_solver: np.ndarray) -> np.ndarray:
    h, w = output_solver.shape
    binary_solver = ndimage.binary_fill_holes(output_solver > 0.5)            # :185
    labeled, nr_objects = ndimage.label(binary_solver)                        # :186
    nb_pixel = [np.sum(labeled == i) for i in range(nr_objects + 1)]
    pixel_order = np.argsort(nb_pixel)
    try:
        return labeled == pixel_order[-2]                                      # :191 second largest label
    except IndexError:
        return np.ones((h, w), dtype=bool)


def _solve(rgb_dev: torch.Tensor, target_dev: torch.Tensor, sigma_spatial, sigma_luma, sigma_chroma) -> np.ndarray:
    soft, _ = _ops.bilateral_solve(rgb_dev, target_dev, sigma_spatial, sigma_luma, sigma_chroma,
                                   confidence=0.999, lam=256.0, a_diag_min=1e-5, cg_tol=1e-5, cg_maxiter=25)   # :162-175
    return soft.cpu().numpy()


def bilateral_solver_output(img, target: np.ndarray, sigma_spatial=16, sigma_luma=16, sigma_chroma=8,
                            device: torch.device = torch.device("cuda:0")):
    assert len(target.shape) == 2, ValueError(f"{len(target.shape)} != 2")
    reference = np.ascontiguousarray(np.array(img))                            # PIL RGB -> u8 [H,W,3]
    rgb = torch.from_numpy(reference).to(device)
    t = np.ascontiguousarray(target)
    t_dev = torch.from_numpy(t if t.dtype == np.uint8 else t.astype(np.double)).to(device)
    output_solver = _solve(rgb, t_dev, sigma_spatial, sigma_luma, sigma_chroma)
    return output_solver, _postprocess(output_solver)


def bilateral_solver_output_from_tensor(x: torch.Tensor, target: torch.Tensor, sigma_spatial=16, sigma_luma=16, sigma_chroma=8):
    """Same result as bilateral_solver_output(convert_tensor_to_pil_image(x), target) (selfmask.py:227-230) without the
    D2H -> PIL -> H2D round trip: x is the normalised image f32 [3,H,W] on the GPU, target u8 [H,W] on the GPU."""
    rgb = _ops.denormalize_u8(x.float().contiguous())                          # utils/utils.py:261-273
    output_solver = _solve(rgb, target.contiguous(), sigma_spatial, sigma_luma, sigma_chroma)
    return output_solver, _postprocess(output_solver)
