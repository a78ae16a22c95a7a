"""Eager vs native-plan replay (one stream, and two half-batch plans on two streams).  usage: plan_bench.py B [iters]"""
import sys, time
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zutis_amd import detgen, plan as zplan
from zutis_amd.engine import ZutisEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
cfg = detgen.VIT_B16
sd = {k: torch.from_numpy(v).to(dev) for k, v in detgen.zutis_state_dict(cfg).items()}
text = torch.from_numpy(detgen.text_embeddings(81, cfg.embed_dim)).to(dev)
x = torch.from_numpy(detgen.images(B, 336, 336, seed=1)).to(dev)


def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / iters * 1e3


eng = ZutisEngine(sd, cfg.patch, cfg.dec_heads)
eng.cross_ksplit = int(os.environ.get("ZH_CROSS_KSPLIT", "2"))     # the drop-in modules' setting (they serve batch-1 loops)
def eager():
    o = eng.forward(x); return eng.predict_semantic(o["patch_tokens"], text, (336, 336))
ms = timeit(eager); print(f"B={B} eager        {ms:.3f} ms  {B / ms * 1e3:.0f} img/s")
p = eng.build_plan(tuple(x.shape), text, (336, 336)); p["x"].copy_(x)
ms = timeit(lambda: eng.run_plan(p)); print(f"B={B} plan ({p['plan'].n} launches) {ms:.3f} ms  {B / ms * 1e3:.0f} img/s")
lab = eager().clone(); eng.run_plan(p); torch.cuda.synchronize(); print("plan == eager:", torch.equal(lab, p["labels"]))
if B >= 2:
    h = B // 2
    ea, eb = ZutisEngine(sd, cfg.patch, cfg.dec_heads), ZutisEngine(sd, cfg.patch, cfg.dec_heads)
    pa, pb = ea.build_plan((h,) + tuple(x.shape[1:]), text, (336, 336)), eb.build_plan((B - h,) + tuple(x.shape[1:]), text, (336, 336))
    pa["x"].copy_(x[:h]); pb["x"].copy_(x[h:])
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    ms = timeit(lambda: zplan.run2(pa["plan"], sa.cuda_stream, pb["plan"], sb.cuda_stream))
    print(f"B={B} plan x2 streams {ms:.3f} ms  {B / ms * 1e3:.0f} img/s")
    print("2-stream == eager:", torch.equal(lab[:h], pa["labels"]) and torch.equal(lab[h:], pb["labels"]))
