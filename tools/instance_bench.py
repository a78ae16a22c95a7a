"""Instance-predict timing (zutis.py:374-470 path) on random-weight outputs (developer tool)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "zutis_amd", "dropin"))
import numpy as np, torch
from zutis_amd import detgen
from networks.zutis import ZUTIS
dev = torch.device("cuda:0")
cfg = detgen.VIT_B16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
net = ZUTIS(categories=[f"c{i}" for i in range(81)], device=dev, text_embeddings=torch.from_numpy(detgen.text_embeddings(81, 512)))
net.load_state_dict({k: torch.from_numpy(v) for k, v in detgen.zutis_state_dict(cfg).items()}, strict=True)
net = net.to(dev).eval()
x = torch.from_numpy(detgen.images(B, 336, 336)).to(dev)
with torch.no_grad():
    out = net(x)
    for nms in ("hard", None):
        net.predict(out, mask_type="instance", size=(336, 336), nms_type=nms)
        torch.cuda.synchronize(); t = time.perf_counter()
        preds = net.predict(out, mask_type="instance", size=(336, 336), nms_type=nms)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        print(f"B={B} nms={nms}: {dt*1e3:.1f} ms total, {len(preds)} predictions, {dt/B*1e3:.1f} ms/image")
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable(); net.predict(out, mask_type="instance", size=(336, 336), nms_type="hard"); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
