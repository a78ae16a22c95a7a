"""Developer: what the cross-attention K / V projections and the cross-attention itself cost the headline step WITH THREE BATCHES IN
FLIGHT — the launch plans are recorded with those calls left out (results are garbage; timing only).  An upper bound for what any
rewrite of that part (round-2 review item 4: absorbing the projections into the attention) could gain: the rewrite has to re-spend
its own MFMA work.     gpurun -- python tools/skip_bound.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zutis_amd import detgen, ops, engine_base
from zutis_amd import plan as zplan
from zutis_amd.engine import ZutisEngine
dev = torch.device("cuda:0")
cfg = detgen.VIT_B16
B, S, n = 32, 336, 81
P = {k: torch.from_numpy(v).to(dev) for k, v in detgen.zutis_state_dict(cfg).items()}
text = torch.from_numpy(detgen.text_embeddings(n, cfg.embed_dim)).to(dev)
x = torch.randn((B, 3, S, S), generator=torch.Generator().manual_seed(1000)).to(dev)
real_gemm, real_attn = engine_base._EngineBase._gemm, ops.attention
skip = set()
real_ln = ops.layernorm
sites = set()
def gemm(self, site, *a, **k):
    sites.add(site)
    if site == "dec_kv" and "kv" in skip: return None
    if ("site:" + site) in skip: return None
    return real_gemm(self, site, *a, **k)
def attention(q, k_, v, o, **kw):
    if "ca" in skip and kw.get("Tk", 0) >= 1024 and kw.get("heads") == 8: return None
    if "enc_attn" in skip and kw.get("heads") == 12: return None
    return real_attn(q, k_, v, o, **kw)
def layernorm(*a, **k):
    if "ln" in skip: return None
    return real_ln(*a, **k)
engine_base._EngineBase._gemm = gemm
ops.attention = attention
ops.layernorm = layernorm
eng = ZutisEngine(P, cfg.patch, cfg.dec_heads, precision="exact")
engs = [eng, eng.fork(), eng.fork()]
for e in engs: e.forward(x)
streams = [torch.cuda.Stream(device=dev) for _ in engs]
def measure(what):
    skip.clear(); skip.update(what)
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
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(7): zplan.run_many(plans, hs)      # 21 steps
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 21)
    return best * 1e3
base = measure(())
print(f"full step, three in flight: {base:.3f} ms ({B / base * 1e3:.0f} images/s)")
for what, name in ((("kv",), "without the two K / V projection GEMMs"), (("ca",), "without the six cross-attention launches"), (("kv", "ca"), "without both")):
    t = measure(what)
    print(f"{name:45s}: {t:.3f} ms  (-{base - t:.3f} ms, {100 * (base - t) / base:.1f} %)")
if "--all" in sys.argv:            # attribution of the in-flight step to kernel families (each left out alone)
    for what in [("ln",), ("enc_attn",)] + [("site:" + st,) for st in sorted(sites)]:
        t = measure(what)
        print(f"without {what[0]:42s}: {t:.3f} ms  (-{base - t:.3f} ms, {100 * (base - t) / base:.1f} %)")
t = measure(())
print(f"full step again: {t:.3f} ms")
