"""Developer: does the PHASE between the three lanes matter?  All lanes replay the same plan; zh_plan_run_multi enqueues them call
by call, so they start in phase (all three at the same kernel).  Here lanes 2 and 3 are first delayed by 1/3 and 2/3 of a step
(torch.cuda._sleep on their streams) — every stream then advances at the same average rate, so the offset persists.
    gpurun -- python tools/stagger_probe.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zutis_amd import detgen, ops
from zutis_amd import plan as zplan
from zutis_amd.engine import ZutisEngine
dev = torch.device("cuda:0")
cfg = detgen.VIT_B16
B, S, n = 32, 336, 81
P = {k: torch.from_numpy(v).to(dev) for k, v in detgen.zutis_state_dict(cfg).items()}
text = torch.from_numpy(detgen.text_embeddings(n, cfg.embed_dim)).to(dev)
x = torch.randn((B, 3, S, S), generator=torch.Generator().manual_seed(1000)).to(dev)
eng = ZutisEngine(P, cfg.patch, cfg.dec_heads, precision="exact")
engs = [eng, eng.fork(), eng.fork()]
for e in engs: e.forward(x)
streams = [torch.cuda.Stream(device=dev) for _ in engs]
plans = []
for e in engs:
    with zplan.Recorder() as rec:
        out = e.forward(x)
        lo = e.semantic_logits_lowres(out["patch_tokens"], text)
        labels = torch.empty((B, S, S), dtype=torch.int64, device=dev)
        ops.upsample_argmax(lo, labels, B, n, lo.shape[2], lo.shape[3], S, S)
    plans.append(rec.build())
hs = [s.cuda_stream for s in streams]
for _ in range(3): zplan.run_many(plans, hs)
torch.cuda.synchronize()
ROUNDS = 20                                           # 60 steps
def run(offsets_ms):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s, off in zip(streams, offsets_ms):
        if off > 0:
            with torch.cuda.stream(s): torch.cuda._sleep(int(off * 1e-3 * 2.0e9))     # ~cycles at ~2 GHz
    for _ in range(ROUNDS): zplan.run_many(plans, hs)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
for name, offs in (("in phase", (0, 0, 0)), ("1/3 step apart", (0, 3.6, 7.2)), ("in phase", (0, 0, 0)), ("1/6 step apart", (0, 1.8, 3.6)), ("1/3 step apart", (0, 3.6, 7.2))):
    t = min(run(offs) for _ in range(2))
    print(f"{name:16s}: {t:8.2f} ms for {3 * ROUNDS} steps = {t / (3 * ROUNDS):.3f} ms/step incl. the one-off offsets ({B * 3 * ROUNDS / t * 1e3:.0f} images/s)")
