"""Config 3 at its own shape and batch (coco20k_eval.py:241-268: batch 1, native-resolution image, forward + instance predict with
hard NMS), through the drop-in module, default precision.  Weights / text / threshold = the config-3 fixture's (so that NMS has
9 categories and ~100 candidates to work on).  Reports ms per image, split into forward and predict."""
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
for prec in ("exact", "fast"):
    net.precision = prec
    for (H, W) in ((480, 640), (427, 640)):
        x = torch.from_numpy(detgen.images(1, H, W, seed=21)).to(dev)
        for _ in range(3):
            out = net(x); preds = net.predict(out, mask_type="instance", threshold=detgen.C3_THRESHOLD, size=(H, W), image_ids=[7], nms_type="hard")
        torch.cuda.synchronize()
        n = 20
        t0 = time.perf_counter()
        for _ in range(n):
            out = net(x)
        torch.cuda.synchronize(); tf = (time.perf_counter() - t0) / n
        t0 = time.perf_counter()
        for _ in range(n):
            preds = net.predict(out, mask_type="instance", threshold=detgen.C3_THRESHOLD, size=(H, W), image_ids=[7], nms_type="hard")
        torch.cuda.synchronize(); tp = (time.perf_counter() - t0) / n
        print(f"[{prec}] {H}x{W}: forward {tf*1e3:.2f} ms + instance predict (hard NMS, {len(preds)} kept of 100) {tp*1e3:.2f} ms = "
              f"{(tf+tp)*1e3:.2f} ms per image ({1/(tf+tp):.0f} images/s)", flush=True)
if "--profile" in sys.argv:
    import cProfile, pstats
    net.precision = "exact"
    x = torch.from_numpy(detgen.images(1, 480, 640, seed=21)).to(dev)
    out = net(x)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(10):
        net.predict(out, mask_type="instance", threshold=detgen.C3_THRESHOLD, size=(480, 640), image_ids=[7], nms_type="hard")
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
