"""Fixed-overhead vs per-slice time of the GEMM kernel: one full round of 256x256 tiles, K swept (developer tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zutis_amd import ops
dev = torch.device("cuda:0")
for (M, N) in [(4096, 4096), (14144, 3072), (14144, 768)]:
    for out_dt in (torch.float16,):
        res = []
        for K in (64, 128, 256, 512, 768, 1024, 2048, 4096):
            A = torch.randn(M, K, device=dev).half(); W = torch.randn(N, K, device=dev).half()
            out = torch.empty(M, N, device=dev, dtype=out_dt)
            for _ in range(3): ops.gemm(A, W, out)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(20): ops.gemm(A, W, out)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
            res.append((K, dt * 1e6))
        print(M, N, " ".join(f"K={k}:{t:.1f}us" for k, t in res))
        (k0, t0), (k1, t1) = res[3], res[-1]
        per_slice = (t1 - t0) / ((k1 - k0) / 32)
        print(f"   per 32-slice {per_slice:.3f} us, extrapolated fixed cost {t0 - per_slice * k0 / 32:.1f} us")
