"""A/B of compile-time variants of the product GEMM (csrc/gemm.hip built with extra -D flags into tools/_abl/), ABBA-timed in one
process on the model's shapes and checked against the first variant.  `--build` compiles (hipcc, no GPU needed)."""
import ctypes as C, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
VARIANTS = {"product": []}                      # plus one variant per -DNAME[=v] argument (e.g. -DZH_X_NODMA, see gemm_kernel.h)
for a in sys.argv[1:]:
    if a.startswith("-D"):
        VARIANTS[a[2:].lower()] = [a]
if "--build" in sys.argv:
    os.makedirs(os.path.join(HERE, "_abl"), exist_ok=True)
    for name, flags in VARIANTS.items():
        out = os.path.join(HERE, "_abl", f"libgemm_{name}.so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared"] + flags +
                              [os.path.join(ROOT, "zutis_amd/csrc/gemm.hip"), os.path.join(ROOT, "zutis_amd/csrc/capi.hip"), "-o", out])
    sys.exit(0)
import torch
dev = torch.device("cuda:0")
vp, l, i = C.c_void_p, C.c_long, C.c_int
names = [n for n in VARIANTS if os.path.exists(os.path.join(HERE, "_abl", f"libgemm_{n}.so"))]
libs = {}
for n in names:
    L = C.CDLL(os.path.join(HERE, "_abl", f"libgemm_{n}.so"))
    L.zh_gemm_f16.restype = i
    L.zh_gemm_f16.argtypes = [vp, l, l, vp, l, l, vp, l, l, i, vp, vp, l, l, i, vp, vp, l, i, i, i, i, i, i, i, i, vp]
    libs[n] = L
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, M, N, K, f16out, res, act in [("qkv", 14144, 2304, 768, 1, 0, 0), ("fc", 14144, 3072, 768, 1, 0, 1), ("out", 14144, 768, 768, 0, 1, 0),
                                        ("proj", 14144, 768, 3072, 0, 1, 0), ("c5fc", 147712, 4096, 1024, 1, 0, 1), ("big", 8192, 8192, 8192, 1, 0, 0)]:
    A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) * 0.03).half()
    out = torch.empty(M, N, device=dev, dtype=torch.float16 if f16out else torch.float32)
    if res: out.normal_()
    b = torch.randn(N, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    fns = {n: (lambda L=libs[n]: L.zh_gemm_f16(A.data_ptr(), K, 0, W.data_ptr(), K, 0, out.data_ptr(), N, 0, f16out, b.data_ptr(),
                                                out.data_ptr() if res else None, N, 0, M if res else 0, None, None, 0, 0, 0, 0, act, M, N, K, 1, s)) for n in names}
    ref = None; diffs = []
    for n in names:
        if res: out.zero_()
        assert fns[n]() == 0, libs[n]
        torch.cuda.synchronize()
        o = out.float().clone()
        if ref is None: ref = o
        else: diffs.append(f"{n}:{float((o - ref).abs().max()):.1e}")
    best = {n: 1e9 for n in names}
    for order in (names, names[::-1], names, names[::-1]):
        for n in order: best[n] = min(best[n], t(fns[n]))
    print(f"{name:5s}", "  ".join(f"{n}:{best[n]:7.1f} us ({2.0 * M * N * K / best[n] / 1e6:4.0f} TF/s)" for n in names), " diff", " ".join(diffs), flush=True)
