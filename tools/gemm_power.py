"""Developer tool: data-dependent (power-limited) GEMM rate.  Same kernel and shape, different operand statistics."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zutis_amd import ops
dev = torch.device("cuda:0")
def mk(kind, M, K):
    if kind == "zeros": return torch.zeros(M, K, device=dev).half()
    if kind == "randn": return torch.randn(M, K, device=dev).half()
    if kind == "relu": return torch.relu(torch.randn(M, K, device=dev)).half()         # post-ReLU activations: half zeros
    if kind == "w0.03": return (torch.randn(M, K, device=dev) * 0.03).half()            # trained-weight scale
    raise ValueError(kind)
for name, M, N, K in (("qkv", 14144, 2304, 768), ("proj", 14144, 768, 3072), ("ffn1b", 56448, 2048, 2048), ("big", 8192, 8192, 8192)):
    for ka, kw in (("zeros", "zeros"), ("randn", "randn"), ("randn", "w0.03"), ("relu", "w0.03")):
        A, W = mk(ka, M, K), mk(kw, N, K)
        out = torch.empty(M, N, device=dev, dtype=torch.float16)
        res = []
        for tile in (None, "5256"):
            if tile: os.environ["ZH_GEMM_TILE"] = tile
            else: os.environ.pop("ZH_GEMM_TILE", None)
            for _ in range(5): ops.gemm(A, W, out)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(20): ops.gemm(A, W, out)
            torch.cuda.synchronize(); res.append(2 * M * N * K / ((time.perf_counter() - t) / 20) / 1e12)
        os.environ.pop("ZH_GEMM_TILE", None)
        Wt = W.t()
        for _ in range(3): torch.mm(A, Wt, out=out)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): torch.mm(A, Wt, out=out)
        torch.cuda.synchronize(); d2 = (time.perf_counter() - t) / 20
        print(f"{name:6s} A={ka:6s} W={kw:6s}: LDS-DMA {res[0]:7.1f}  reg-staged {res[1]:7.1f}  hipblaslt {2*M*N*K/d2/1e12:7.1f} TF")
