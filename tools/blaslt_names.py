"""Developer tool: which hipBLASLt kernels torch.mm picks for the model's fp16 GEMM shapes (run under rocprofv3 --kernel-trace)."""
import torch
dev = torch.device("cuda:0")
for M, N, K in [(14144, 2304, 768), (14144, 768, 768), (14144, 3072, 768), (14144, 768, 3072), (147712, 3072, 1024), (147712, 1024, 4096), (8192, 8192, 8192)]:
    A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) * 0.03).half()
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    for _ in range(3):
        torch.mm(A, W.t(), out=out)
    torch.cuda.synchronize()
