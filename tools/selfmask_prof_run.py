"""Target of the rocprofv3 passes for the pseudo-label path (profiles/r02_selfmask_*): SelfMask (DINO ViT-S/8, 512x683, T = 5505)
+ bilateral solver + nearest resize on the device, batch 1 and batch 4."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zutis_amd import detgen, pseudo_masks
from zutis_amd.engine import SelfMaskEngine
dev = torch.device("cuda:0")
H, W = 512, 683
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
PREC = sys.argv[2] if len(sys.argv) > 2 else "exact"
eng = SelfMaskEngine({k: torch.from_numpy(v).to(dev) for k, v in detgen.selfmask_state_dict().items()}, precision=PREC)
import bench
x = bench.natural_images(B, H, W, dev, seed=7)       # natural colour statistics, as bench.py's pseudo_labels object (round 6)
for _ in range(2):
    pseudo_masks.pseudo_masks_batch(eng, x, [(480, 640)] * B, True)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    pseudo_masks.pseudo_masks_batch(eng, x, [(480, 640)] * B, True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 5
print(f"[{PREC}] SelfMask + solver + resize, batch {B} @ {H}x{W}: {dt*1e3:.2f} ms per batch, {B/dt:.1f} images/s (device side)")
