"""Developer tool: S independent batches of B images in flight on S streams (native plan replay).  usage: B S [iters]"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zutis_amd import detgen, plan as zplan
from zutis_amd.engine import ZutisEngine
B = int(sys.argv[1]); S = int(sys.argv[2]); iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda:0"); cfg = detgen.VIT_B16
sd = {k: torch.from_numpy(v).to(dev) for k, v in detgen.zutis_state_dict(cfg).items()}
text = torch.from_numpy(detgen.text_embeddings(81, cfg.embed_dim)).to(dev)
engs = [ZutisEngine(sd, cfg.patch, cfg.dec_heads) for _ in range(S)]
plans = [e.build_plan((B, 3, 336, 336), text, (336, 336)) for e in engs]
for i, p in enumerate(plans): p["x"].copy_(torch.from_numpy(detgen.images(B, 336, 336, seed=i)).to(dev))
streams = [torch.cuda.Stream() for _ in range(S)]
torch.cuda.synchronize()
def go(): zplan.run_many([p["plan"] for p in plans], [s.cuda_stream for s in streams])
for _ in range(2): go()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(iters): go()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / iters
print(f"B={B} x {S} streams: {dt*1e3:.3f} ms per round, {dt*1e3/S:.3f} ms per batch, {B*S/dt:.0f} img/s")
