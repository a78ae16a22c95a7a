"""Developer: config-3 forward time (batch 1, 480x640 / 427x640, exact) against the cross-attention key split of the drop-in module."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
for (H, W) in ((480, 640), (336, 336)):
    x = torch.from_numpy(detgen.images(1, H, W, seed=21)).to(dev)
    ref = None
    for ks in (1, 2, 4, 8, 12, 16):
        net.cross_attention_key_split = ks; net._engine = None
        for _ in range(3): out = net(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): out = net(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
        mp = out["mask_proposals"].float()
        if ref is None: ref = mp.clone()
        print(f"{H}x{W} key split {ks:2d}: forward {dt*1e3:.3f} ms   max |mask proposal - split 1| {float((mp - ref).abs().max()):.2e}", flush=True)
