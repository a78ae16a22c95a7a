"""Developer tool: pure-Python cost of one eager forward (launches recorded, not executed) + cProfile top entries."""
import sys, os, time, cProfile, pstats, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zutis_amd import detgen, plan as zplan
from zutis_amd.engine import ZutisEngine
dev = torch.device("cuda:0"); cfg = detgen.VIT_B16
sd = {k: torch.from_numpy(v).to(dev) for k, v in detgen.zutis_state_dict(cfg).items()}
text = torch.from_numpy(detgen.text_embeddings(81, cfg.embed_dim)).to(dev)
eng = ZutisEngine(sd, cfg.patch, cfg.dec_heads)
x = torch.from_numpy(detgen.images(1, 336, 336, seed=1)).to(dev)
def step():
    o = eng.forward(x); return eng.predict_semantic(o["patch_tokens"], text, (336, 336))
step(); torch.cuda.synchronize()
with zplan.Recorder() as rec:
    t = time.perf_counter()
    for _ in range(20): step()
    dt = (time.perf_counter() - t) / 20
n = len(rec.calls) // 20
print(f"python-only cost of one forward+predict: {dt*1e3:.3f} ms for {n} launches = {dt/n*1e6:.2f} us per launch")
with zplan.Recorder() as rec:
    pr = cProfile.Profile(); pr.enable()
    for _ in range(20): step()
    pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
