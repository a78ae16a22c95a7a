"""Tile comparison for the fp32-out + bias + in-place residual epilogue (the out_proj / c_proj form).  ZH_GEMM_TILE is read once
per process: run once per tile code, e.g.  for t in auto 64 128 192 256; do ZH_GEMM_TILE=$t python tools/gemm_tiles_res.py; done"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if os.environ.get("ZH_GEMM_TILE") == "auto":
    del os.environ["ZH_GEMM_TILE"]
from zutis_amd import ops
dev = torch.device("cuda:0")
out = []
for name, M, N, K in [("out", 14144, 768, 768), ("proj", 14144, 768, 3072), ("c4out", 8200, 768, 768), ("c4proj", 8200, 768, 3072)]:
    A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) * 0.03).half()
    X = torch.randn(M, N, device=dev); bias = torch.randn(N, device=dev)
    for _ in range(3): ops.gemm(A, W, X, bias=bias, residual=X)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): ops.gemm(A, W, X, bias=bias, residual=X)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    out.append(f"{name}:{us:6.1f} ({2.0 * M * N * K / us / 1e6:4.0f} TF/s)")
print(f"{os.environ.get('ZH_GEMM_TILE', 'auto'):>5}", "  ".join(out))
