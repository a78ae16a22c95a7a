"""One long GEMM (4096 x 4096 x K=32768) for counter-based diagnosis of the steady-state loop."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zutis_amd import ops
dev = torch.device("cuda:0")
M = N = 4096; K = 32768
A = torch.randn(M, K, device=dev).half(); W = torch.randn(N, K, device=dev).half(); out = torch.empty(M, N, device=dev, dtype=torch.float16)
for _ in range(3): ops.gemm(A, W, out)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): ops.gemm(A, W, out)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
print(f"K={K}: {dt*1e3:.3f} ms  {2*M*N*K/dt/1e12:.1f} TF/s")
