"""Developer tool (round 5): the plain fp16 GEMM (zh_gemm_f16) under forced tile codes, interleaved in ONE process (rounds of
[variant A, variant B, ...] per shape, median and min), model-shaped operands (A ~ N(0,1), W ~ N(0, 0.03^2)), next to the vendor
library (torch.mm -> hipBLASLt) on the same data; results checked against the auto tile bit for bit (same K order).

    python tools/gemm_f16_tiles.py [rounds] [tile codes ...]      # default codes: 0 (auto) 256 192
(profiles/r05_gemm_f16_tiles.txt was taken with codes 0 6256 6192: the 64-k two-slot loop of the round-5 experiment tree, since removed)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from zutis_amd import _lib, ops

dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
codes = [int(a) for a in sys.argv[2:]] or [0, 256, 192]
L = _lib.load()
shapes = [("qkv", 14144, 2304, 768, "f16", 0), ("out", 14144, 768, 768, "f32r", 0), ("fc", 14144, 3072, 768, "f16", 1),
          ("proj", 14144, 768, 3072, "f32r", 0), ("kv", 56448, 4608, 256, "f16", 0),
          ("L.qkv", 147712, 3072, 1024, "f16", 0), ("L.out", 147712, 1024, 1024, "f32r", 0), ("L.fc", 147712, 4096, 1024, "f16", 1),
          ("L.proj", 147712, 1024, 4096, "f32r", 0), ("sq4k", 4096, 4096, 4096, "f16", 0), ("sq8k", 8192, 8192, 8192, "f16", 0)]


def timeit(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for name, M, N, K, kind, act in shapes:
    g = torch.Generator(device=dev).manual_seed(1)
    A = torch.randn((M, K), generator=g, device=dev).half()
    W = (torch.randn((N, K), generator=g, device=dev) * 0.03).half()
    bias = torch.randn((N,), generator=g, device=dev) * 0.02
    if kind == "f16":
        out = torch.empty((M, N), dtype=torch.float16, device=dev)
        res = None
    else:
        out = torch.empty((M, N), dtype=torch.float32, device=dev)
        res = torch.randn((M, N), generator=g, device=dev)
    flops = 2.0 * M * N * K
    n_in = max(3, int(4e-3 / (flops / 0.9e15)))        # ~4 ms of launches per timing

    def run(code):
        L.zh_dev_set_gemm_overrides(0, code, 0)
        ops.gemm(A, W, out, bias=bias, act=act, residual=res, res_rows=M if res is not None else 0)

    ref = None
    same = {}
    for c in codes:
        run(c); torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        else:
            same[c] = bool(torch.equal(out, ref))
    Wt = W.t()
    o16 = torch.empty((M, N), dtype=torch.float16, device=dev)
    ts = {c: [] for c in codes}
    tb = []
    for r in range(rounds):
        for c in codes:
            run(c)
            ts[c].append(timeit(lambda: run(c), n_in))
        torch.mm(A, Wt, out=o16)
        tb.append(timeit(lambda: torch.mm(A, Wt, out=o16), n_in))
    L.zh_dev_set_gemm_overrides(0, 0, 0)
    line = f"{name:7s} {M:6d}x{N:4d}x{K:4d} {kind:4s}"
    for c in codes:
        med, mn = float(np.median(ts[c])), min(ts[c])
        line += f" | {c:4d}: {med * 1e6:7.1f} us (min {mn * 1e6:7.1f}) {flops / med / 1e12:6.0f} TF" + ("" if c == codes[0] else (" ==" if same[c] else " !="))
    medb = float(np.median(tb))
    line += f" | blaslt(f16 out, no epilogue): {medb * 1e6:7.1f} us {flops / medb / 1e12:6.0f} TF"
    print(line, flush=True)
    del A, W, out, res, ref, o16
    torch.cuda.empty_cache()
