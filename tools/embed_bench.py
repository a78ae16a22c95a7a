"""Developer tool: config-5 throughput — CLIP ViT-L/14@336 image-embedding extraction (ClipImageEncoder), random-init weights."""
import sys, os, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zutis_amd import detgen
from zutis_amd.engine import ClipImageEncoder
dev = torch.device("cuda:0")
D, L, p, g, E = 1024, 24, 14, 24, 768
def w(name, shape, std, mean=0.0): return torch.from_numpy(detgen.det_normal(name, shape, std, mean, 5)).to(dev)
P = {"visual.class_embedding": w("cls", (D,), D ** -0.5), "visual.positional_embedding": w("pos", (g * g + 1, D), D ** -0.5),
     "visual.proj": w("proj", (D, E), D ** -0.5), "visual.conv1.weight": w("conv", (D, 3, p, p), (3 * p * p) ** -0.5)}
for ln in ("ln_pre", "ln_post"):
    P[f"visual.{ln}.weight"] = w(ln + "w", (D,), 0.1, 1.0); P[f"visual.{ln}.bias"] = w(ln + "b", (D,), 0.1)
for i in range(L):
    q = f"visual.transformer.resblocks.{i}."
    P[q + "attn.in_proj_weight"] = w(q + "a", (3 * D, D), D ** -0.5); P[q + "attn.in_proj_bias"] = w(q + "ab", (3 * D,), 0.02)
    P[q + "attn.out_proj.weight"] = w(q + "o", (D, D), D ** -0.5 * (2 * L) ** -0.5); P[q + "attn.out_proj.bias"] = w(q + "ob", (D,), 0.02)
    P[q + "mlp.c_fc.weight"] = w(q + "f", (4 * D, D), (2 * D) ** -0.5); P[q + "mlp.c_fc.bias"] = w(q + "fb", (4 * D,), 0.02)
    P[q + "mlp.c_proj.weight"] = w(q + "p", (D, 4 * D), D ** -0.5 * (2 * L) ** -0.5); P[q + "mlp.c_proj.bias"] = w(q + "pb", (D,), 0.02)
    for ln in ("ln_1", "ln_2"):
        P[q + ln + ".weight"] = w(q + ln + "w", (D,), 0.1, 1.0); P[q + ln + ".bias"] = w(q + ln + "b", (D,), 0.1)
enc = ClipImageEncoder(P, p, prefix="visual.")
T = g * g + 1
flop = L * (2 * T * (D * 3 * D + D * D + 2 * D * 4 * D) + 4 * T * T * D) + 2 * g * g * 3 * p * p * D
for B in (32, 64, 128):
    x = torch.randn(B, 3, 336, 336, device=dev)
    for _ in range(2): e = enc.encode_image(x)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): e = enc.encode_image(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    print(f"ViT-L/14@336 B={B}: {B/dt:.0f} images/s ({dt*1e3:.1f} ms), {B*flop/dt/1e12:.0f} TFLOP/s, |e|={e.norm(dim=1).mean().item():.4f}")
