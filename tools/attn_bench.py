"""Attention micro-benchmark over the model's shapes (developer tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zutis_amd import ops
dev = torch.device("cuda:0")
out = []
for name, B, H, dh, Tq, Tk in [("enc", 32, 12, 64, 442, 442), ("cross", 32, 8, 96, 100, 1764), ("self", 32, 8, 96, 100, 100), ("c4enc", 8, 12, 64, 1025, 1025),
                              ("c5enc", 256, 16, 64, 577, 577), ("selfmask", 1, 6, 64, 5505, 5505)]:
    D = H * dh
    q = torch.randn(B, Tq, D, device=dev).half(); k = torch.randn(B, Tk, D, device=dev).half(); v = torch.randn(B, Tk, D, device=dev).half()
    o = torch.empty(B, Tq, D, device=dev, dtype=torch.float16)
    run = lambda: ops.attention(q, k, v, o, batch=B, heads=H, Tq=Tq, Tk=Tk, head_dim=dh, ldq=D, ldk=D, ldv=D, ldo=D, strideQ=Tq * D, strideK=Tk * D,
                                strideV=Tk * D, strideO=Tq * D)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    out.append(f"{name}: {us:7.1f} us ({4.0 * B * H * Tq * Tk * dh / us / 1e6:5.0f} TF/s)")
print("  ".join(out))
