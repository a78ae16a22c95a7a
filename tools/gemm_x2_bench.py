"""f16x2 (fp16-valued weight, planeW = 0) against f16x3 on the shapes of BASELINE config 5 (CLIP ViT-L/14@336, 256 images) and of
the C2 encoder.  TFLOP/s are ALGORITHMIC (2*M*N*K).  --tile T forces a tile code for the x2 runs (5122 = two-slot 256x256)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zutis_amd import ops, _lib
from zutis_amd.ops import Act
dev = torch.device("cuda:0")
tiles = [int(a) for a in sys.argv[1:]] or [0]
shapes = [(147712, 3072, 1024, "L qkv", "split"), (147712, 1024, 1024, "L out", "f32"), (147712, 4096, 1024, "L fc", "split"),
          (147712, 1024, 4096, "L proj", "f32"), (14144, 2304, 768, "B qkv", "split"), (14144, 768, 768, "B out", "f32"),
          (14144, 3072, 768, "B fc", "split"), (14144, 768, 3072, "B proj", "f32")]
L = _lib.load(raw=True)
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for M, N, K, name, kind in shapes:
    A32 = torch.randn(M, K, device=dev); W32 = (torch.randn(N, K, device=dev) * 0.03).half().float()
    A = Act.empty((M, K), True, dev); ops.cast_f16(A32, A, M, K)
    W2, W3 = ops.split_weight(W32), ops.split_weight(W32, allow_x2=False)
    res = torch.randn(M, N, device=dev) if kind == "f32" else None
    out = torch.empty(M, N, device=dev) if kind == "f32" else Act.empty((M, N), True, dev)
    act = ops.ACT_QUICKGELU if "fc" in name else ops.ACT_NONE
    L.zh_dev_set_gemm_overrides(0, 0, 0)
    d3 = t(lambda: ops.gemm_x3(A, W3, out, residual=res, act=act))
    line = f"{name:7s} {M}x{N}x{K} out={kind:5s} x3 {d3*1e6:8.1f} us {2*M*N*K/d3/1e12:6.1f} TF/s"
    for tl in tiles:
        L.zh_dev_set_gemm_overrides(0, tl, 0)
        d2 = t(lambda: ops.gemm_x3(A, W2, out, residual=res, act=act))
        line += f" | x2[{tl}] {d2*1e6:8.1f} us {2*M*N*K/d2/1e12:6.1f} TF/s ({d3/d2:.2f}x)"
    L.zh_dev_set_gemm_overrides(0, 0, 0)
    print(line, flush=True)
