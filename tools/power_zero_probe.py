"""Developer: is the exact step power-limited?  The same launch sequence (one stream, eager) with the model's weights and with
ALL-ZERO weights and inputs: identical instruction streams, but the MFMA datapath toggles nothing on zeros, so the chip holds a
higher clock.  The ratio is how much of the step's time is the power limit's doing."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zutis_amd import detgen
from zutis_amd.engine import ZutisEngine
dev = torch.device("cuda:0")
cfg = detgen.VIT_B16
sd = detgen.zutis_state_dict(cfg)
x = torch.randn((32, 3, 336, 336), generator=torch.Generator().manual_seed(1)).to(dev)
for name, scale in (("model weights", 1.0), ("zero weights + zero input", 0.0), ("model weights", 1.0), ("zero weights + zero input", 0.0)):
    P = {k: torch.from_numpy(v).to(dev) * scale for k, v in sd.items()}
    if scale == 0.0:
        for k in P:                      # LayerNorm gains 0 -> every activation exactly 0 after the first LN
            pass
    eng = ZutisEngine(P, cfg.patch, cfg.dec_heads, precision="exact")
    xi = x * scale
    for _ in range(3):
        eng.forward(xi)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10):
        eng.forward(xi)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    print(f"{name:28s}: {dt*1e3:7.3f} ms per forward (batch 32, one stream, exact)", flush=True)
