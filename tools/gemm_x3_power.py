"""Developer tool: what bounds the f16x3 GEMM — structure or power?  Same kernel and shape with all-zero operands (the MFMA
datapath toggles nothing: the chip holds its top clock) and with model-shaped operands (A ~ N(0,1), W ~ N(0, 0.03^2)).
Also prints the fp16-operand kernel on the same shapes.  TFLOP/s are algorithmic; x3 MFMA issue = 3x."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zutis_amd import ops
from zutis_amd.ops import Act
dev = torch.device("cuda:0")
shapes = [(14144, 2304, 768, "qkv", "split"), (14144, 768, 768, "out", "f32"), (14144, 3072, 768, "fc", "split"), (14144, 768, 3072, "proj", "f32"),
          (56448, 4608, 256, "kv-all", "split"), (3200, 768, 768, "dec768", "split"), (3200, 2048, 768, "dec l1", "split"), (3200, 768, 768, "dec768f", "f32"), (3200, 768, 2048, "dec l2", "f32"), (8192, 8192, 8192, "8k", "f32")]
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
tag = os.environ.get("ZUTIS_HIP_LIB", "product")
if os.environ.get("ZH_X3_TILE"):                      # forced x3 tile code (64 | 192 | 256 | 512) for every shape
    from zutis_amd import _lib
    _lib.load(raw=True).zh_dev_set_gemm_overrides(0, int(os.environ["ZH_X3_TILE"]), 0)
    tag += "-tile" + os.environ["ZH_X3_TILE"]
for M, N, K, name, kind in shapes:
    row = f"[{os.path.basename(tag)}] {name:7s} {M}x{N}x{K} out={kind:5s}"
    for data in ("zeros", "model"):
        if data == "zeros":
            A = Act(torch.zeros((2, M, K), dtype=torch.float16, device=dev)); W = Act(torch.zeros((2, N, K), dtype=torch.float16, device=dev))
        else:
            A32 = torch.randn(M, K, device=dev); W32 = torch.randn(N, K, device=dev) * 0.03
            A = Act.empty((M, K), True, dev); ops.cast_f16(A32, A, M, K)
            W = ops.split_weight(W32)
        out = torch.empty(M, N, device=dev) if kind == "f32" else Act.empty((M, N), kind == "split", dev)
        res = None if kind != "f32" else out
        dt3 = t(lambda: ops.gemm_x3(A, W, out, residual=res))
        row += f" | {data}: x3 {dt3*1e6:7.1f} us {2*M*N*K/dt3/1e12:6.1f} TF (pipe {3*2*M*N*K/dt3/1e15:4.2f} PF)"
        if "NO" not in tag and "tile" not in tag:
            o2 = out if kind == "f32" else Act.empty((M, N), False, dev)
            dt1 = t(lambda: ops.gemm(A.hi, W.hi, o2, residual=res))
            row += f" f16 {dt1*1e6:6.1f} us {2*M*N*K/dt1/1e12:6.1f} TF"
    print(row, flush=True)
