import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zutis_amd import detgen, ops
from zutis_amd.engine import ZutisEngine
from zutis_amd import engine_base
dev = torch.device("cuda:0")
cfg = detgen.VIT_B16
P = {k: torch.from_numpy(v).to(dev) for k, v in detgen.zutis_state_dict(cfg).items()}
x = torch.from_numpy(detgen.images(4, 336, 336, seed=4)).to(dev)
eng = ZutisEngine(P, cfg.patch, cfg.dec_heads)
full = {k: v.clone() for k, v in eng.forward(x).items()}
orig = ops.attention
for name, patch in (("as is", None), ("no enc key split", "noks"), ("no splitk", "nosk"), ("no skinny", "nosk2")):
    if patch == "noks":
        def att(*a, **kw):
            if kw.get("Tq") == kw.get("Tk"): kw["ksplit"] = 1; kw["workspace"] = None
            return orig(*a, **kw)
        ops.attention = att
    elif patch == "nosk":
        ops.attention = orig
        engine_base._EngineBase.SPLITK_MAX_ROWS = 0
    one = eng.forward(x[:1].contiguous())
    print(name, "tokens", float((one["patch_tokens"][0] - full["patch_tokens"][0]).abs().max()), "masks", float((one["mask_proposals"][0] - full["mask_proposals"][0]).abs().max()))
engine_base._EngineBase.SPLITK_MAX_ROWS = 2048
from oracle import zutis_ref as O
with torch.no_grad():
    ref = O.zutis_forward(O.to_torch_params(detgen.zutis_state_dict(cfg)), x[:1].cpu(), cfg.patch, cfg.dec_heads)
for nm, o in (("B=4 run", {k: v[:1] for k, v in full.items()}), ("B=1 run", eng.forward(x[:1].contiguous()))):
    print(nm, "vs oracle: tokens", float((o["patch_tokens"].cpu() - ref["patch_tokens"]).abs().max()), "masks", float((o["mask_proposals"].cpu() - ref["mask_proposals"]).abs().max()))
