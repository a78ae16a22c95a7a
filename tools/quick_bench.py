"""Quick per-stage timing of the HIP engine at the C2 workload (developer tool; bench.py is the contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zutis_amd import detgen
from zutis_amd.engine import ZutisEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
cfg = detgen.VIT_B16
P = {k: torch.from_numpy(v).to(dev) for k, v in detgen.zutis_state_dict(cfg).items()}
eng = ZutisEngine(P, cfg.patch, cfg.dec_heads)
x = torch.randn(B, 3, 336, 336, device=dev)
text = torch.from_numpy(detgen.text_embeddings(81, 512)).to(dev)
for _ in range(3):
    out = eng.forward(x); lab = eng.predict_semantic(out["patch_tokens"], text, (336, 336))
torch.cuda.synchronize()
def timeit(fn, n=10):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
te = timeit(lambda: eng.encode(x))
tf = timeit(lambda: eng.forward(x))
tp = timeit(lambda: eng.predict_semantic(out["patch_tokens"], text, (336, 336)))
tg = timeit(lambda: eng.forward_graphed(x)) if B <= 8 else float("nan")
print(f"B={B} graphed forward {tg:.2f} ms")
print(f"B={B} encode {te:.2f} ms  forward {tf:.2f} ms  predict {tp:.2f} ms  -> {B / (tf + tp) * 1e3:.1f} img/s")
flops = B * 124.4e9
print(f"forward ~{flops / tf / 1e9:.1f} TFLOP/s")
