"""Bilateral solver throughput at the SelfMask size (512x683), one image per call and batched (zh_bilateral_solve_batch).
Algorithmic bytes per image: N*(3+1+8+4*4) + V*250*(25 CG + 11 bistochastisation iterations)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from zutis_amd import ops, detgen
dev = torch.device("cuda:0")
H, W = 512, 683
yy, xx = np.mgrid[:H, :W]
for B in (1, 2, 4, 8, 16, 32):
    rgb = torch.from_numpy(np.stack([detgen.selfmask_like_rgb(H, W, seed=3 + i) for i in range(B)])).to(dev)
    tg = torch.from_numpy(np.stack([(((yy - 250) ** 2 + (xx - 300 - 3 * i) ** 2) < 150 ** 2).astype(np.uint8) for i in range(B)])).to(dev)
    soft, stats = ops.bilateral_solve(rgb, tg)
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        ops.bilateral_solve(rgb, tg)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    V = stats[:, 0].float().mean().item()
    byts = H * W * (3 + 1 + 8 + 16) + V * 250 * 36
    print(f"B={B:2d}: {dt*1e3:7.3f} ms per call, {dt/B*1e3:6.3f} ms per image, V~{V:.0f}, iters {stats[:,1].tolist()[:4]}, "
          f"algorithmic {byts/1e6:.1f} MB/image -> {byts*B/dt/1e12:.3f} TB/s", flush=True)
