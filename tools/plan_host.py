import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zutis_amd import detgen
from zutis_amd.engine import ZutisEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0"); cfg = detgen.VIT_B16
sd = {k: torch.from_numpy(v).to(dev) for k, v in detgen.zutis_state_dict(cfg).items()}
text = torch.from_numpy(detgen.text_embeddings(81, cfg.embed_dim)).to(dev)
eng = ZutisEngine(sd, cfg.patch, cfg.dec_heads)
p = eng.build_plan((B, 3, 336, 336), text, (336, 336))
for _ in range(3): eng.run_plan(p)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); eng.run_plan(p); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"B={B} enqueue {1e3*(t1-t0):.3f} ms, total {1e3*(t2-t0):.3f} ms, launches {p['plan'].n}")
# 10 replays back-to-back
t0 = time.perf_counter()
for _ in range(10): eng.run_plan(p)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"10x: enqueue {1e2*(t1-t0):.3f} ms/replay, total {1e2*(t2-t0):.3f} ms/replay")
