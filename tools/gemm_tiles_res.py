"""Tile comparison for the fp32-out + bias + in-place residual epilogue (the out_proj / c_proj form)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zutis_amd import ops
dev = torch.device("cuda:0")
for name, M, N, K in [("out", 14144, 768, 768), ("proj", 14144, 768, 3072), ("c4out", 8200, 768, 768), ("c4proj", 8200, 768, 3072)]:
    A = torch.randn(M, K, device=dev).half(); W = torch.randn(N, K, device=dev).half()
    X = torch.randn(M, N, device=dev); bias = torch.randn(N, device=dev)
    res = {}
    for tile in ("auto", "128", "192", "256"):
        if tile == "auto": os.environ.pop("ZH_GEMM_TILE", None)
        else: os.environ["ZH_GEMM_TILE"] = tile
        for _ in range(3): ops.gemm(A, W, X, bias=bias, residual=X)
        ts = []
        for r in range(3):
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(20): ops.gemm(A, W, X, bias=bias, residual=X)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t) / 20 * 1e6)
        res[tile] = min(ts)
    print(f"{name:8s}", " ".join(f"{k}:{v:7.1f}" for k, v in res.items()))
