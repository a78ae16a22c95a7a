"""Target of the rocprofv3 pass for config 3's forward at its own shape and batch (profiles/r03_c3_forward_kernel_stats.csv):
the drop-in module, batch 1, 480x640, 21 forwards."""
import sys, os
ROOT = "/root/repo" if os.path.exists("/root/repo/zutis_amd") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "zutis_amd", "dropin"))
import numpy as np, torch
from zutis_amd import detgen
from networks.zutis import ZUTIS
dev = torch.device("cuda:0")
cfg = detgen.VIT_B16
g = np.load(os.path.join(ROOT, "tests", "golden", "c3_vitb16.npz"))
net = ZUTIS(categories=[f"c{i}" for i in range(81)], clip_arch="ViT-B/16", device=dev, text_embeddings=torch.from_numpy(g["text"]))
net.load_state_dict({k: torch.from_numpy(v) for k, v in detgen.c3_state_dict(cfg).items()}, strict=True)
net = net.to(dev).eval().requires_grad_(False)
x = torch.from_numpy(detgen.images(1, 480, 640, seed=21)).to(dev)
out = net(x)
for _ in range(20):
    out = net(x)
torch.cuda.synchronize()
