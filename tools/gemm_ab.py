"""A/B two GEMM source variants in ONE process on ONE device (cdna guide rule 24).  Builds tools/ab/gemm_<v>.hip into
private .so files and interleaves timed rounds.  python tools/gemm_ab.py v3 v4"""
import ctypes as C, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from zutis_amd import _lib
variants = sys.argv[1:] or ["v3", "v4"]
libs = {}
for v in variants:
    so = f"/tmp/libgemm_{v}.so"
    src = os.path.join(ROOT, "tools", "ab", f"gemm_{v}.hip")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-I", os.path.join(ROOT, "zutis_amd", "csrc"),
                           src, os.path.join(ROOT, "zutis_amd", "csrc", "capi.hip"), "-o", so])
    L = C.CDLL(so); L.zh_gemm_f16.restype = C.c_int; L.zh_gemm_f16.argtypes = _lib._SIGS["zh_gemm_f16"][1]; libs[v] = L
dev = torch.device("cuda:0")
shapes = [("qkv", 14144, 2304, 768), ("out", 14144, 768, 768), ("fc", 14144, 3072, 768), ("proj", 14144, 768, 3072), ("kv", 56448, 4608, 768), ("sq", 4096, 4096, 4096)]
st = torch.cuda.current_stream().cuda_stream
for name, M, N, K in shapes:
    A = torch.randn(M, K, device=dev).half(); W = torch.randn(N, K, device=dev).half(); out = torch.empty(M, N, device=dev, dtype=torch.float16)
    def run(L, n):
        for _ in range(n):
            rc = L.zh_gemm_f16(A.data_ptr(), K, 0, W.data_ptr(), K, 0, out.data_ptr(), N, 0, 1, None, None, 0, 0, 0, 0, M, N, K, 1, st); assert rc == 0
    res = {v: [] for v in variants}
    for v in variants: run(libs[v], 3)
    for rnd in range(7):
        for v in variants:
            torch.cuda.synchronize(); t = time.perf_counter(); run(libs[v], 10); torch.cuda.synchronize()
            res[v].append((time.perf_counter() - t) / 10 * 1e6)
    print(name, " ".join(f"{v}: med {sorted(r)[3]:.1f} min {min(r):.1f} us" for v, r in res.items()))
