"""Compare forced tile shapes in one process (ZH_GEMM_TILE is read per call)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zutis_amd import ops
dev = torch.device("cuda:0")
shapes = [("qkv", 14144, 2304, 768), ("out", 14144, 768, 768), ("fc", 14144, 3072, 768), ("proj", 14144, 768, 3072), ("kv", 56448, 4608, 768),
          ("ffn1a", 56448, 256, 768), ("ts", 56448, 512, 768), ("dec", 3200, 768, 768), ("dec_ff1", 3200, 2048, 768), ("dec_ff2", 3200, 768, 2048)]
for name, M, N, K in shapes:
    A = torch.randn(M, K, device=dev).half(); W = torch.randn(N, K, device=dev).half(); out = torch.empty(M, N, device=dev, dtype=torch.float16)
    res = {}
    for tile in ("auto", "128", "192", "256"):
        if tile == "auto": os.environ.pop("ZH_GEMM_TILE", None)
        else: os.environ["ZH_GEMM_TILE"] = tile
        for _ in range(3): ops.gemm(A, W, out)
        ts = []
        for r in range(3):
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(20): ops.gemm(A, W, out)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t) / 20 * 1e6)
        res[tile] = min(ts)
    print(f"{name:8s}", " ".join(f"{k}:{v:7.1f}" for k, v in res.items()))
