"""Target of the rocprofv3 passes for the bilateral solver (profiles/r02_bilateral_*): B images of 512x683, a few calls."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from zutis_amd import ops, detgen
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H, W = 512, 683
yy, xx = np.mgrid[:H, :W]
rgb = torch.from_numpy(np.stack([detgen.selfmask_like_rgb(H, W, seed=3 + i) for i in range(B)])).to(dev)
tg = torch.from_numpy(np.stack([(((yy - 250) ** 2 + (xx - 300 - 3 * i) ** 2) < 150 ** 2).astype(np.uint8) for i in range(B)])).to(dev)
for _ in range(5):
    ops.bilateral_solve(rgb, tg)
torch.cuda.synchronize()
