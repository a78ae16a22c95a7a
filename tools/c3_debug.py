"""Developer: per-decoder-layer mask-proposal error of the HIP engine against the CPU oracle on the config-3 fixture model."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from zutis_amd import detgen
from zutis_amd.engine import ZutisEngine
from oracle import zutis_ref as O
dev = torch.device("cuda:0")
cfg = detgen.VIT_B16
H, W = 427, 640
x = torch.from_numpy(detgen.images(1, H, W, seed=21))
from zutis_amd import _lib
if len(sys.argv) > 1:
    _lib.load(raw=True).zh_dev_set_gemm_overrides(0, int(sys.argv[1]), 0)
    print("forced x3 tile", sys.argv[1])
for name, sd in (("plain", detgen.zutis_state_dict(cfg)),):
    with torch.no_grad():
        ref = O.zutis_forward(O.to_torch_params(sd), x, cfg.patch, cfg.dec_heads)
    for prec in ("exact",):
        eng = ZutisEngine({k: torch.from_numpy(v).to(dev) for k, v in sd.items()}, cfg.patch, cfg.dec_heads, precision=prec)
        out = eng.forward(x.to(dev))
        mp = out["mask_proposals"].cpu()
        print(name, prec, "per-layer max |mask err|:", [f"{float((mp[:, l] - ref['mask_proposals'][:, l]).abs().max()):.2e}" for l in range(mp.shape[1])],
              "patch tokens", f"{float((out['patch_tokens'].cpu() - ref['patch_tokens']).abs().max()):.2e}", flush=True)
