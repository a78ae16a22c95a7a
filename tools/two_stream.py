"""Experiment: split the batch over two HIP streams (two engines sharing the parameters) so one half's kernel tails /
small decoder kernels overlap the other half's big GEMMs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zutis_amd import detgen, ops
from zutis_amd.engine import ZutisEngine
dev = torch.device("cuda:0")
cfg = detgen.VIT_B16
P = {k: torch.from_numpy(v).to(dev) for k, v in detgen.zutis_state_dict(cfg).items()}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
x = torch.randn(B, 3, 336, 336, device=dev)
text = torch.from_numpy(detgen.text_embeddings(81, 512)).to(dev)
engs = [ZutisEngine(P, cfg.patch, cfg.dec_heads) for _ in range(NS)]
streams = [torch.cuda.Stream() for _ in range(NS)]
xs = list(x.chunk(NS))
def step():
    outs = []
    for e, s, xi in zip(engs, streams, xs):
        with torch.cuda.stream(s):
            o = e.forward(xi); outs.append(e.predict_semantic(o["patch_tokens"], text, (336, 336)))
    return outs
single = ZutisEngine(P, cfg.patch, cfg.dec_heads)
def step1():
    o = single.forward(x); return single.predict_semantic(o["patch_tokens"], text, (336, 336))
for f, name in ((step1, "1 stream"), (step, f"{NS} streams")):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    print(f"{name}: {dt*1e3:.2f} ms/step  {B/dt:.0f} img/s")
