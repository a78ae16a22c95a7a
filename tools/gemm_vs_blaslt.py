"""Developer tool: our GEMM vs the vendor library (torch.mm -> hipBLASLt) on the model's shapes, to locate the ceiling."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zutis_amd import ops
dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
shapes = [("qkv", 14144, 2304, 768), ("out", 14144, 768, 768), ("fc", 14144, 3072, 768), ("proj", 14144, 768, 3072),
          ("kv", 56448, 1536, 768), ("ffn1a", 56448, 2048, 768), ("ffn1b", 56448, 2048, 2048), ("ffn1c", 56448, 768, 2048), ("ts", 56448, 512, 768),
          ("dec_ff1", 3200, 2048, 768), ("sq", 4096, 4096, 4096), ("big", 8192, 8192, 8192)]
def t(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters
for name, M, N, K in shapes:
    A = torch.randn(M, K, device=dev).half(); W = torch.randn(N, K, device=dev).half()
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    d1 = t(lambda: ops.gemm(A, W, out))
    Wt = W.t()
    d2 = t(lambda: torch.mm(A, Wt, out=out))
    print(f"{name:8s} M={M:6d} N={N:5d} K={K:5d}  ours {d1*1e6:8.1f} us {2*M*N*K/d1/1e12:7.1f} TF | hipblaslt {d2*1e6:8.1f} us {2*M*N*K/d2/1e12:7.1f} TF")
