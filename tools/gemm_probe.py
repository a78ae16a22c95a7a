"""Developer tool: per-block phase timestamps of the GEMM (needs tools/ab/libgemm_probe.so built with -DZH_GEMM_PROBE).
Phases: start -> prologue landed -> K loop done -> stores drained.  Times in us (100 MHz wall clock)."""
import sys, os, ctypes as C
import torch, numpy as np
here = os.path.dirname(os.path.abspath(__file__))
variant = os.environ.get("PROBE_LIB", "libgemm_probe.so")
print("#### variant", variant)
lib = C.CDLL(os.path.join(here, "ab", variant))
vp, l, i = C.c_void_p, C.c_long, C.c_int
lib.zh_gemm_f16.restype = i
lib.zh_gemm_f16.argtypes = [vp, l, l, vp, l, l, vp, l, l, i, vp, vp, l, l, i, i, i, i, i, i, vp]
lib.zh_gemm_set_probe.argtypes = [vp]
dev = torch.device("cuda:0")
shapes = [("qkv", 14144, 2304, 768, 1), ("proj", 14144, 768, 3072, 0)]
for name, M, N, K, of16 in shapes:
    A = torch.randn(M, K, device=dev).half(); W = torch.randn(N, K, device=dev).half()
    out = torch.empty(M, N, device=dev, dtype=torch.float16 if of16 else torch.float32)
    res = torch.randn(M, N, device=dev) if not of16 else None
    probe = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
    def run():
        rc = lib.zh_gemm_f16(A.data_ptr(), K, 0, W.data_ptr(), K, 0, out.data_ptr(), N, 0, of16, None,
                             res.data_ptr() if res is not None else None, N, 0, M if res is not None else 0, 0, M, N, K, 1,
                             torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    lib.zh_gemm_set_probe(None)
    for _ in range(5): run()
    lib.zh_gemm_set_probe(probe.data_ptr())
    run(); torch.cuda.synchronize()
    lib.zh_gemm_set_probe(None)
    raw = probe.cpu().numpy().reshape(-1, 8)
    raw = raw[raw[:, 0] > 0]
    mhz = (raw[:, 6] - raw[:, 5]) / ((raw[:, 2] - raw[:, 1]) / 100.0)
    print(f"   shader clock during the K loop: {mhz.mean():.0f} MHz (min {mhz.min():.0f}, max {mhz.max():.0f})")
    t = raw[:, :4].astype(np.float64) / 100.0   # us
    t0 = t[:, 0].min()
    order = np.argsort(t[:, 0])
    t = t[order] - t0
    nb = len(t)
    print(f"== {name} M={M} N={N} K={K} blocks={nb}  kernel span {t[:,3].max():.1f} us")
    pro, loop, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    first = t[:, 0] < 1.0
    for lab, sel in (("round-1 blocks", first), ("later blocks", ~first)):
        if sel.sum() == 0: continue
        print(f"  {lab:15s} n={sel.sum():4d} start {t[sel,0].mean():6.2f}  prologue {pro[sel].mean():5.2f}  kloop {loop[sel].mean():6.2f} (min {loop[sel].min():.2f} max {loop[sel].max():.2f})  epilogue {epi[sel].mean():5.2f} (max {epi[sel].max():.2f})  end {t[sel,3].mean():6.2f}")
