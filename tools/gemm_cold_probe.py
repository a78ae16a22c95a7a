"""Developer: the decoder's small f16x3 GEMMs hot (same launch repeated) vs cold (512 MB written between launches: L2 and the
Infinity Cache hold nothing of the operands), to separate "cold weights" from launch gaps in the 19 us isolated / ~30 us in-chain gap."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zutis_amd import ops
from zutis_amd.ops import Act
dev = torch.device("cuda:0")
junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for M, N, K, name, kind in [(3200, 768, 768, "dec 768x768 f32+res", "f32"), (3200, 768, 768, "dec 768x768 split", "split"), (3200, 2048, 768, "dec l1", "split"),
                            (3200, 768, 2048, "dec l2", "f32"), (3200, 2304, 768, "dec qkv", "split")]:
    A32 = torch.randn(M, K, device=dev); W32 = torch.randn(N, K, device=dev) * 0.03
    A = Act.empty((M, K), True, dev); ops.cast_f16(A32, A, M, K)
    W = ops.split_weight(W32)
    out = torch.empty(M, N, device=dev) if kind == "f32" else Act.empty((M, N), True, dev)
    res = out if kind == "f32" else None
    run = lambda: ops.gemm_x3(A, W, out, residual=res)
    for _ in range(5): run()
    def timed(cold, weights_only=False):
        ts = []
        for _ in range(8):
            if cold:
                junk.fill_(1)
                if weights_only:                                   # activations warm again (the producer kernel just wrote them)
                    A.t.add_(0); (out if kind == "f32" else out.t).add_(0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        return ts[len(ts) // 2]
    print(f"{name:22s} {M}x{N}x{K}: hot {timed(False):6.1f} us   all operands cold {timed(True):6.1f} us   only the weights cold {timed(True, True):6.1f} us", flush=True)
