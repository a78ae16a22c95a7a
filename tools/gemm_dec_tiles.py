"""Decoder-shaped GEMMs (M = B*Q = 3200 rows) under a forced tile code (ZH_GEMM_TILE is read once per process)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zutis_amd import ops
dev = torch.device("cuda:0")
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
res = []
for (M, N, K, f16out, resid) in [(3200, 1536, 768, 1, 0), (3200, 768, 768, 1, 0), (3200, 768, 768, 0, 1), (3200, 2048, 768, 1, 0), (3200, 768, 2048, 0, 1), (800, 768, 768, 0, 1), (800, 2048, 768, 1, 0)]:
    A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) * 0.03).half()
    out = torch.empty(M, N, device=dev, dtype=torch.float16 if f16out else torch.float32)
    R = torch.randn(M, N, device=dev) if resid else None
    b = torch.randn(N, device=dev)
    us = t(lambda: ops.gemm(A, W, out, bias=b, residual=R))
    res.append(f"{M}x{N}x{K}{'r' if resid else ' '}:{us:6.1f}")
print(os.environ.get("ZH_GEMM_TILE", "auto"), " ".join(res))
