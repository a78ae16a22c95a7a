"""Developer tool (round 5): persistent big plain-fp16 GEMM tiles (zh_dev_set_gemm_persist) against one workgroup per tile, interleaved
in one process on the model's shapes (model-shaped operands), bitwise-compared; then the C2 `fast` step and the C5 `fast` step A/B."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from zutis_amd import _lib, ops
dev = torch.device("cuda:0")
L = _lib.load()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
shapes = [("qkv", 14144, 2304, 768, "f16", 0), ("fc", 14144, 3072, 768, "f16", 1), ("kv", 56448, 4608, 256, "f16", 0),
          ("proj", 14144, 768, 3072, "f32r", 0), ("c4.fc", 8200, 3072, 768, "f16", 1),
          ("L.qkv", 147712, 3072, 1024, "f16", 0), ("L.fc", 147712, 4096, 1024, "f16", 1), ("L.proj", 147712, 1024, 4096, "f32r", 0)]
def timeit(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for name, M, N, K, kind, act in shapes:
    g = torch.Generator(device=dev).manual_seed(1)
    A = torch.randn((M, K), generator=g, device=dev).half(); W = (torch.randn((N, K), generator=g, device=dev) * 0.03).half()
    bias = torch.randn((N,), generator=g, device=dev) * 0.02
    out = torch.empty((M, N), dtype=torch.float16 if kind == "f16" else torch.float32, device=dev)
    res = torch.randn((M, N), generator=g, device=dev) if kind != "f16" else None
    flops = 2.0 * M * N * K
    n_in = max(3, int(4e-3 / (flops / 0.9e15)))
    def run(pers):
        L.zh_dev_set_gemm_persist(pers)
        ops.gemm(A, W, out, bias=bias, act=act, residual=res, res_rows=M if res is not None else 0)
    run(0); torch.cuda.synchronize(); ref = out.clone()
    run(256); torch.cuda.synchronize(); same = bool(torch.equal(out, ref))
    ts = {0: [], 256: []}
    for r in range(rounds):
        for c in (0, 256):
            run(c); ts[c].append(timeit(lambda: run(c), n_in))
    m0, m1 = float(np.median(ts[0])), float(np.median(ts[256]))
    print(f"{name:7s} {M:6d}x{N:4d}x{K:4d} {kind:4s} | one WG per tile {m0*1e6:8.1f} us {flops/m0/1e12:6.0f} TF | persistent {m1*1e6:8.1f} us {flops/m1/1e12:6.0f} TF | {100*(m0/m1-1):+5.1f} % | bitwise {'==' if same else '!='}", flush=True)
L.zh_dev_set_gemm_persist(256)
