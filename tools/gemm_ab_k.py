"""A/B variants over a K sweep at one-round shapes (fixed cost vs per-slice)."""
import ctypes as C, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from zutis_amd import _lib
variants = sys.argv[1:]
libs = {}
for v in variants:
    so = f"/tmp/libgemm_{v}.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-I", os.path.join(ROOT, "zutis_amd", "csrc"),
                           os.path.join(ROOT, "tools", "ab", f"gemm_{v}.hip"), os.path.join(ROOT, "zutis_amd", "csrc", "capi.hip"), "-o", so])
    L = C.CDLL(so); L.zh_gemm_f16.restype = C.c_int; L.zh_gemm_f16.argtypes = _lib._SIGS["zh_gemm_f16"][1]; libs[v] = L
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
for (M, N) in [(4096, 4096), (14144, 768)]:
    for K in (64, 256, 768, 3072):
        A = torch.randn(M, K, device=dev).half(); W = torch.randn(N, K, device=dev).half(); out = torch.empty(M, N, device=dev, dtype=torch.float16)
        def run(L, n):
            for _ in range(n):
                assert L.zh_gemm_f16(A.data_ptr(), K, 0, W.data_ptr(), K, 0, out.data_ptr(), N, 0, 1, None, None, 0, 0, 0, 0, M, N, K, 1, st) == 0
        res = {v: [] for v in variants}
        for v in variants: run(libs[v], 3)
        for rnd in range(5):
            for v in variants:
                torch.cuda.synchronize(); t = time.perf_counter(); run(libs[v], 20); torch.cuda.synchronize()
                res[v].append((time.perf_counter() - t) / 20 * 1e6)
        print(M, N, K, " ".join(f"{v}: {sorted(r)[2]:.1f}us" for v, r in res.items()))
