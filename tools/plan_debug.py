import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zutis_amd import detgen, plan as zplan
from zutis_amd.engine import ZutisEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0"); cfg = detgen.VIT_B16
sd = {k: torch.from_numpy(v).to(dev) for k, v in detgen.zutis_state_dict(cfg).items()}
text = torch.from_numpy(detgen.text_embeddings(81, cfg.embed_dim)).to(dev)
x = torch.from_numpy(detgen.images(B, 336, 336, seed=1)).to(dev)
eng = ZutisEngine(sd, cfg.patch, cfg.dec_heads)
o1 = eng.forward(x); l1 = eng.predict_semantic(o1["patch_tokens"], text, (336, 336))
o2 = eng.forward(x); l2 = eng.predict_semantic(o2["patch_tokens"], text, (336, 336))
print("eager twice:", torch.equal(o1["mask_proposals"], o2["mask_proposals"]), torch.equal(o1["patch_tokens"], o2["patch_tokens"]), torch.equal(l1, l2))
p = eng.build_plan(tuple(x.shape), text, (336, 336))
for r in range(3):
    out, lab = eng.run_plan(p, x); torch.cuda.synchronize()
    print("plan rep", r, torch.equal(out["mask_proposals"], o1["mask_proposals"]), torch.equal(out["patch_tokens"], o1["patch_tokens"]), torch.equal(lab, l1),
          (out["mask_proposals"] - o1["mask_proposals"]).abs().max().item(), (lab != l1).sum().item())
o3 = eng.forward(x); l3 = eng.predict_semantic(o3["patch_tokens"], text, (336, 336))
print("eager after:", torch.equal(o3["mask_proposals"], o1["mask_proposals"]), torch.equal(l3, l1))
if B >= 2:
    h = B // 2
    ea, eb = ZutisEngine(sd, cfg.patch, cfg.dec_heads), ZutisEngine(sd, cfg.patch, cfg.dec_heads)
    pa, pb = ea.build_plan((h,) + tuple(x.shape[1:]), text, (336, 336)), eb.build_plan((B - h,) + tuple(x.shape[1:]), text, (336, 336))
    ea.run_plan(pa, x[:h].contiguous()); eb.run_plan(pb, x[h:].contiguous()); torch.cuda.synchronize()
    print("halves sequential:", torch.equal(pa["labels"], l1[:h]), torch.equal(pb["labels"], l1[h:]), torch.equal(pa["out"]["mask_proposals"], o1["mask_proposals"][:h]))
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream(); torch.cuda.synchronize()
    zplan.run2(pa["plan"], sa.cuda_stream, pb["plan"], sb.cuda_stream); torch.cuda.synchronize()
    print("halves concurrent:", torch.equal(pa["labels"], l1[:h]), torch.equal(pb["labels"], l1[h:]), torch.equal(pa["out"]["mask_proposals"], o1["mask_proposals"][:h]),
          (pa["labels"] != l1[:h]).sum().item(), (pb["labels"] != l1[h:]).sum().item())
