"""A/B two attention source variants in one process (tools/ab/attn_<v>.hip)."""
import ctypes as C, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from zutis_amd import _lib
variants = sys.argv[1:] or ["old", "new"]
libs = {}
for v in variants:
    so = f"/tmp/libattn_{v}.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-I", os.path.join(ROOT, "zutis_amd", "csrc"),
                           os.path.join(ROOT, "tools", "ab", f"attn_{v}.hip"), os.path.join(ROOT, "zutis_amd", "csrc", "capi.hip"), "-o", so])
    L = C.CDLL(so); L.zh_attention_f16.restype = C.c_int; L.zh_attention_f16.argtypes = _lib._SIGS["zh_attention_f16"][1]; libs[v] = L
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
for name, B, H, dh, Tq, Tk in [("enc", 32, 12, 64, 442, 442), ("cross", 32, 8, 96, 100, 1764), ("self", 32, 8, 96, 100, 100), ("selfmask", 1, 6, 64, 5505, 5505)]:
    D = H * dh
    q = torch.randn(B, Tq, D, device=dev).half(); k = torch.randn(B, Tk, D, device=dev).half(); v_ = torch.randn(B, Tk, D, device=dev).half()
    o = torch.empty(B, Tq, D, device=dev, dtype=torch.float16)
    def run(L, n):
        for _ in range(n):
            assert L.zh_attention_f16(q.data_ptr(), D, Tq * D, k.data_ptr(), D, Tk * D, v_.data_ptr(), D, Tk * D, o.data_ptr(), D, Tq * D, B, H, Tq, Tk, dh, dh ** -0.5, st) == 0
    res = {v: [] for v in variants}
    for v in variants: run(libs[v], 3)
    for r in range(5):
        for v in variants:
            torch.cuda.synchronize(); t = time.perf_counter(); run(libs[v], 20); torch.cuda.synchronize(); res[v].append((time.perf_counter() - t) / 20 * 1e6)
    print(name, " ".join(f"{v}: med {sorted(r)[2]:.1f} us" for v, r in res.items()))
