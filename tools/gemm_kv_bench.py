"""The composed decoder K / V projections (engine._decoder_kv): [B*M, 256] x [L*D, 256]^T -> fp16 [B*M, L*D] (520 MB written
per launch at the bench shape), with and without the `pos @ Wk^T` tables (accumulator preload), per tile code (ZH_GEMM_TILE is read once per
process: run once per code), next to a plain 520 MB device fill as the write-bandwidth reference."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zutis_amd import ops
from zutis_amd.ops import Act
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B, h2, w2, K, N = 32, 42, 42, 256, 4608
M = B * h2 * w2
A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) * 0.03).half()
out = torch.empty(M, N, device=dev, dtype=torch.float16)
b = torch.randn(N, device=dev); Ty = torch.randn(h2, N, device=dev); Tx = torch.randn(w2, N, device=dev)
kt = dict(pos=(Ty, Tx))
res = [f"fill:{t(lambda: out.fill_(1.0)):6.1f}"]
res.append(f"V:{t(lambda: ops.gemm(A, W, out, bias=b)):6.1f}")
res.append(f"K:{t(lambda: ops.gemm(A, W, out, bias=b, **kt)):6.1f}")
kh = dict(pos=(Ty.half(), Tx.half()))
res.append(f"K(f16 tables):{t(lambda: ops.gemm(A, W, out, bias=b, **kh)):6.1f}")
if "--x3" in sys.argv:
    A3 = Act(torch.stack([A, (A * 2 ** -11)]).contiguous()); W3 = ops.split_weight(W.float())
    o1, o2 = Act.empty((M, N), False, dev), Act.empty((M, N), True, dev)
    res.append(f"x3 V f16:{t(lambda: ops.gemm_x3(A3, W3, o1, bias=b)):6.1f}")
    res.append(f"x3 K f16:{t(lambda: ops.gemm_x3(A3, W3, o1, bias=b, **kt)):6.1f}")
    res.append(f"x3 V pair:{t(lambda: ops.gemm_x3(A3, W3, o2, bias=b)):6.1f}")
    res.append(f"x3 K pair:{t(lambda: ops.gemm_x3(A3, W3, o2, bias=b, **kt)):6.1f}")
print(os.environ.get("ZH_GEMM_TILE", "auto"), " ".join(res), "us;", f"{M * N * 2 / 1e6:.0f} MB out")
