"""x3 (split-pair) GEMM throughput on the model's shapes, next to the fp16-operand kernel.  TFLOP/s are ALGORITHMIC
(2*M*N*K): the x3 kernel issues three MFMAs per product, so its MFMA-pipe rate is 3x the printed figure."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zutis_amd import ops
from zutis_amd.ops import Act
dev = torch.device("cuda:0")
shapes = [(14144, 2304, 768, "qkv"), (14144, 768, 768, "out"), (14144, 3072, 768, "fc"), (14144, 768, 3072, "proj"),
          (56448, 4608, 768, "kv-all"), (56448, 256, 768, "ffn1.0"), (56448, 768, 256, "ffn1.2"), (56448, 512, 768, "textproj"),
          (3200, 2048, 768, "dec l1"), (8192, 8192, 8192, "8k")]
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for M, N, K, name in shapes:
    A32 = torch.randn(M, K, device=dev); W32 = torch.randn(N, K, device=dev) * 0.03
    A = Act.empty((M, K), True, dev); ops.cast_f16(A32, A, M, K)
    W = ops.split_weight(W32)
    for kind, out in (("f32", torch.empty(M, N, device=dev)), ("f16", Act.empty((M, N), False, dev)), ("split", Act.empty((M, N), True, dev))):
        dt3 = t(lambda: ops.gemm_x3(A, W, out))
        line = f"{name:9s} {M}x{N}x{K} out={kind:5s} x3 {dt3*1e6:8.1f} us {2*M*N*K/dt3/1e12:7.1f} TF/s"
        if kind != "split" and K % 64 == 0:
            o2 = out.hi if isinstance(out, Act) else out
            dt1 = t(lambda: ops.gemm(A.hi, W.hi, o2))
            line += f" | f16 {dt1*1e6:8.1f} us {2*M*N*K/dt1/1e12:7.1f} TF/s  ratio {dt3/dt1:.2f}"
        print(line, flush=True)
