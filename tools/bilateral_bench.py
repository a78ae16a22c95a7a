"""Bilateral solver: GPU (zh_bilateral_solve) vs the NumPy/SciPy oracle on a SelfMask-sized image (developer tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from zutis_amd import ops, detgen
from oracle import bilateral_ref as B
dev = torch.device("cuda:0")
for (h, w) in [(512, 683), (512, 1131)]:
    rgb = detgen.selfmask_like_rgb(h, w, seed=3)
    yy, xx = np.mgrid[:h, :w]
    target = (((yy - h / 2) ** 2 + (xx - w / 2) ** 2) < (0.3 * h) ** 2).astype(np.uint8)
    r, t = torch.from_numpy(rgb).to(dev), torch.from_numpy(target).to(dev)
    for _ in range(3): soft, stats = ops.bilateral_solve(r, t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): soft, stats = ops.bilateral_solve(r, t)
    torch.cuda.synchronize(); gpu = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter(); ref, _ = B.bilateral_solver_output(rgb, target); cpu = time.perf_counter() - t0
    V, its = stats.cpu().tolist()
    print(f"{h}x{w}: V={V} cg_iters={its} GPU {gpu*1e3:.3f} ms  oracle(CPU, incl. post-processing) {cpu*1e3:.0f} ms  max|diff| {np.abs(soft.cpu().numpy()-ref).max():.2e}")
