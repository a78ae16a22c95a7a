import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zutis_amd import detgen
from oracle import zutis_ref as O
cfg = detgen.VIT_B16
P = O.to_torch_params(detgen.zutis_state_dict(cfg))
x = torch.from_numpy(detgen.images(2, 336, 336))
print("cpu_count", os.cpu_count())
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    with torch.no_grad():
        O.zutis_forward(P, x[:1], cfg.patch, cfg.dec_heads)
        t = time.perf_counter(); O.zutis_forward(P, x, cfg.patch, cfg.dec_heads); dt = time.perf_counter() - t
    print(nt, "threads:", round(2 / dt, 3), "img/s")
