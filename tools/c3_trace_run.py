"""rocprofv3 target: config 3 (batch 1, 480x640) through the drop-in module: N forwards (hipGraph replay when `graph` is given,
else eager) each followed by the instance predict.  usage: c3_trace_run.py [graph] [n]"""
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
if "graph" in sys.argv:
    net.use_hip_graph = True
n = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 10
x = torch.from_numpy(detgen.images(1, 480, 640, seed=21)).to(dev)
for _ in range(n):
    out = net(x)
    if "nopredict" not in sys.argv:
        net.predict(out, mask_type="instance", threshold=detgen.C3_THRESHOLD, size=(480, 640), image_ids=[7], nms_type="hard")
torch.cuda.synchronize()
