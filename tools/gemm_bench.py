"""GEMM micro-benchmark over the model's shapes (developer tool).  python tools/gemm_bench.py [iters]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zutis_amd import ops
dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
shapes = [("qkv", 14144, 2304, 768), ("out", 14144, 768, 768), ("fc", 14144, 3072, 768), ("proj", 14144, 768, 3072),
          ("kv", 56448, 4608, 768), ("ffn1a", 56448, 256, 768), ("ffn1c", 56448, 768, 256), ("ts", 56448, 512, 768),
          ("dec_qk", 3200, 1536, 768), ("dec_ff1", 3200, 2048, 768), ("dec_ff2", 3200, 768, 2048), ("sq", 4096, 4096, 4096)]
for name, M, N, K in shapes:
    A = torch.randn(M, K, device=dev).half(); W = torch.randn(N, K, device=dev).half()
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    for _ in range(3): ops.gemm(A, W, out)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(iters): ops.gemm(A, W, out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / iters
    print(f"{name:8s} M={M:6d} N={N:5d} K={K:5d}  {dt*1e6:8.1f} us  {2*M*N*K/dt/1e12:7.1f} TF/s")
