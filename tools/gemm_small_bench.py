"""Batch-1 GEMM shapes (config 3: T = 1201 encoder tokens, Q = 100 decoder queries) under forced tile codes and split-K
(K split expressed as a batched GEMM over K slabs: batch = S, strideA = strideW = K / S, strideC = M * N), with the weights
rotated through 40 buffers per shape so that they come from HBM as in the model (its 0.5 GB of packed weights do not stay in
the 256 MiB Infinity Cache).  usage: gemm_small_bench.py [enc|dec|all]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zutis_amd import ops, _lib
from zutis_amd.ops import Act
dev = torch.device("cuda:0")
L = _lib.load(raw=True)
NW = 40


def t(fn, n=40):
    """us per launch of n launches replayed from a hipGraph (no host time per launch: the eager Python loop costs ~10 us each,
    more than most of these kernels)."""
    for i in range(4): fn(i)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(0)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n): fn(i)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * n) * 1e3


def bench(M, N, K, name, tiles, splits=(1,), out="pair", act=ops.ACT_NONE, resid=False, NW=NW):
    A32 = torch.randn(M, K, device=dev)
    A = Act.empty((M, K), True, dev); ops.cast_f16(A32, A, M, K)
    Ws = [ops.split_weight(torch.randn(N, K, device=dev) * 0.03) for _ in range(NW)]
    b = torch.randn(N, device=dev)
    R = torch.randn(M, N, device=dev) if resid else None
    ref = (A32.double() @ (Ws[0].t[0].double() + Ws[0].t[1].double()).t() * Ws[0].out_scale).float()
    line = f"{name:8s} {M}x{N}x{K}"
    for S in splits:
        for tile in tiles:
            _lib.check(L.zh_dev_set_gemm_overrides(0, tile, 0))
            try:
                if S == 1:
                    o = Act.empty((M, N), True, dev) if out == "pair" else torch.empty(M, N, device=dev)
                    fn = lambda i: ops.gemm_x3(A, Ws[i % NW], o, bias=b, act=act, residual=R)
                    fn(0); torch.cuda.synchronize()
                    got = (o.t[0].float() + o.t[1].float()) if out == "pair" else o
                    err = float((got - (ops_act(ref + b, act) + (R if resid else 0))).abs().max())
                else:
                    o = torch.empty(S, M, N, device=dev)
                    fn = lambda i: ops.gemm_x3(A, Ws[i % NW], o, M=M, N=N, K=K // S, lda=K, ldw=K, ldc=N, batch=S, strideA=K // S, strideW=K // S, strideC=M * N)
                    fn(0); torch.cuda.synchronize()
                    err = float((o.sum(0) - ref).abs().max())
                us = t(fn)
                line += f" | S{S} t{tile}: {us:5.1f}us e{err:.0e}"
            except Exception as e:
                line += f" | S{S} t{tile}: ERR {str(e)[:40]}"
            finally:
                L.zh_dev_set_gemm_overrides(0, 0, 0)
    print(line, flush=True)


def ops_act(x, act):
    if act == ops.ACT_QUICKGELU: return x * torch.sigmoid(1.702 * x)
    if act == ops.ACT_RELU: return torch.relu(x)
    return x


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    T = 1201
    ENC_TILES = (0, 96, 3064, 1288, 6496)
    if which in ("enc", "all"):
        bench(T, 2304, 768, "qkv", ENC_TILES)
        bench(T, 3072, 768, "fc", ENC_TILES, act=ops.ACT_QUICKGELU)
        bench(T, 768, 768, "out", ENC_TILES, out="f32", resid=True)
        bench(T, 768, 3072, "proj", ENC_TILES, out="f32", resid=True)
        bench(T, 768, 768, "out/S", (0, 1288), splits=(2, 4), out="f32")
        bench(T, 768, 3072, "proj/S", (0, 1288, 6496), splits=(2, 3, 4, 6), out="f32")
        bench(442, 2304, 768, "qkv336", (0, 3064, 6496))
        bench(442, 3072, 768, "fc336", (0, 3064, 6496), act=ops.ACT_QUICKGELU)
        bench(442, 768, 3072, "proj336", (0, 3064), out="f32", resid=True)
        bench(442, 768, 3072, "proj336/S", (0, 3064), splits=(2, 4, 8), out="f32")
    if which == "abl":        # K-loop ablations (ZUTIS_HIP_LIB = a -DZH_X3_NO* variant library): timing only, results are garbage
        bench(T, 2304, 768, "qkv", (96, 1288, 64, 3064))
        bench(T, 3072, 768, "fc", (64, 1288, 3064), act=ops.ACT_QUICKGELU)
    if which == "warm":       # where do the weights come from: 40 buffers (HBM, as in the model), 4 (Infinity Cache), 1 (L2 / Infinity Cache)
        for nw in (40, 4, 1):
            bench(T, 2304, 768, f"qkv/w{nw}", (96, 1288), NW=nw)
            bench(T, 3072, 768, f"fc/w{nw}", (64, 1288), act=ops.ACT_QUICKGELU, NW=nw)
            bench(T, 768, 3072, f"proj/w{nw}", (1288,), splits=(4,), out="f32", NW=nw)
            bench(100, 768, 768, f"dec/w{nw}", (0,), out="f32", resid=True, NW=nw)
            bench(100, 768, 2048, f"l2/w{nw}", (0,), out="f32", resid=True, NW=nw)
    if which == "dec32":      # the decoder's GEMMs at the headline's batch (32 images x 100 queries = 3200 rows) under the tile candidates
        DT = (0, 64, 96, 1288, 6496, 192, 256)
        bench(3200, 768, 768, "dec o", DT, out="f32", resid=True)
        bench(3200, 2304, 768, "dec qkv", DT)
        bench(3200, 768, 768, "dec q", DT)
        bench(3200, 2048, 768, "dec l1", DT, act=ops.ACT_RELU)
        bench(3200, 768, 2048, "dec l2", DT, out="f32", resid=True)
        bench(3200, 768, 2048, "dec l2/S", (0, 96, 1288), splits=(2, 4), out="f32")
        bench(19200, 256, 768, "ffn2.0", (0, 96, 1288, 192, 256), act=ops.ACT_RELU)
        bench(19200, 768, 256, "ffn2.2", (0, 96, 1288, 192, 256), out="f32")
    if which == "k64":        # round 5: 64-k slices on a ring that keeps two or more slices in flight (3064 = 64 x 64 / four slots, 6464 = 128 x 64 / three)
        bench(T, 2304, 768, "qkv", (96, 6496, 7096, 1288, 7128))
        bench(T, 3072, 768, "fc", (1288, 7128, 7096), act=ops.ACT_QUICKGELU)
        bench(T, 768, 3072, "proj/S4", (1288, 7128, 7096), splits=(4,), out="f32")
        bench(3200, 768, 768, "dec o", (0, 96, 7096), out="f32", resid=True)
        bench(3200, 2304, 768, "dec qkv", (0, 7096, 7128))
        bench(3200, 768, 768, "dec q", (0, 96, 7096))
        if len(sys.argv) > 2: sys.exit(0)
        bench(T, 1536, 768, "n1536", (64, 6464, 96, 6496, 1288))          # beside their 32-k forms (3066 = 64 x 64 / six slots, 64 = 128 x 64 / three) and the two-slot 64-k tile (6496)
        bench(T, 768, 768, "out", (3066, 3064), out="f32", resid=True)
        bench(T, 768, 768, "out pair", (3066, 3064))
        bench(442, 2304, 768, "qkv336", (3066, 3064, 6496))
        bench(442, 3072, 768, "fc336", (3066, 3064, 6496), act=ops.ACT_QUICKGELU)
        bench(T, 768, 3072, "proj/S4", (1288, 6464, 6496), splits=(4,), out="f32")
    if which == "k64f":       # round 5, the plain-fp16 kernel (`fast`): one-round tiles on 64-k slices (7032 / 7096 / 7128) beside the 32-k tiles (3064 / 64 / 128)
        def benchf(M, N, K, name, tiles, f16out=True, act=ops.ACT_NONE, resid=False):
            A = torch.randn(M, K, device=dev).half()
            Ws = [(torch.randn(N, K, device=dev) * 0.03).half() for _ in range(NW)]
            b = torch.randn(N, device=dev)
            R = torch.randn(M, N, device=dev) if resid else None
            ref = ops_act(A.float() @ Ws[0].float().t() + b, act) + (R if resid else 0)
            line = f"{name:8s} {M}x{N}x{K}"
            for tile in tiles:
                _lib.check(L.zh_dev_set_gemm_overrides(0, tile, 0))
                try:
                    o = torch.empty(M, N, device=dev, dtype=torch.float16 if f16out else torch.float32)
                    fn = lambda i: ops.gemm(A, Ws[i % NW], o, bias=b, residual=R, act=act)
                    fn(0); torch.cuda.synchronize()
                    err = float((o.float() - ref).abs().max())
                    line += f" | t{tile}: {t(fn):5.1f}us e{err:.0e}"
                except Exception as e:
                    line += f" | t{tile}: ERR {str(e)[:40]}"
                finally:
                    L.zh_dev_set_gemm_overrides(0, 0, 0)
            print(line, flush=True)
        DT = (64, 128, 7096, 7128)
        benchf(3200, 768, 768, "dec o", DT, f16out=False, resid=True)
        benchf(3200, 768, 768, "dec q", DT)
        benchf(3200, 2304, 768, "dec qkv", DT)
        benchf(3200, 2048, 768, "dec l1", DT, act=ops.ACT_RELU)
        benchf(3200, 768, 2048, "dec l2", DT, f16out=False, resid=True)
        ET = (128, 7096, 7128)
        benchf(T, 2304, 768, "qkv", ET)
        benchf(T, 3072, 768, "fc", ET, act=ops.ACT_QUICKGELU)
        benchf(T, 768, 768, "out", (64, 3064, 7032), f16out=False, resid=True)
        benchf(T, 768, 3072, "proj", (64, 3064, 7032), f16out=False, resid=True)
        benchf(442, 2304, 768, "qkv336", (3064, 7032))
        benchf(442, 768, 3072, "proj336", (3064, 7032), f16out=False, resid=True)
    if which == "pmc64":      # counters of the 32-k (96) and 64-k (6496) forms of the one-image QKV tile
        def t(fn, n=12):
            for i in range(n): fn(i)
            torch.cuda.synchronize(); return 0.0
        bench(T, 2304, 768, "qkv", (96, 6496))
    if which == "pmc":        # rocprofv3 --pmc target: a few eager launches of the batch-1 QKV / fc / proj shapes, cold weights
        def t(fn, n=12):
            for i in range(n): fn(i)
            torch.cuda.synchronize(); return 0.0
        bench(T, 2304, 768, "qkv", (0,))
        bench(T, 3072, 768, "fc", (0, 1288), act=ops.ACT_QUICKGELU)
        bench(T, 768, 3072, "proj", (0,), out="f32", resid=True)
    if which in ("dec", "all"):
        for M in (100, 200, 400):
            bench(M, 2304, 768, "sa_qkv", (0, 3064, 32) if M > 128 else (0, 3064))
            bench(M, 768, 768, "proj", (0, 3064, 32) if M > 128 else (0, 3064), out="f32", resid=True)
            bench(M, 2048, 768, "l1", (0, 3064, 32) if M > 128 else (0, 3064), act=ops.ACT_RELU)
            bench(M, 768, 2048, "l2", (0, 3064, 32) if M > 128 else (0, 3064), out="f32", resid=True)
        bench(600, 256, 768, "ffn2.0", (0, 3064, 32), act=ops.ACT_RELU)
        bench(600, 256, 256, "ffn2.1", (0, 3064, 32), act=ops.ACT_RELU)
        bench(600, 768, 256, "ffn2.2", (0, 3064, 32), out="f32")
        bench(20, 384, 384, "sm", (0, 3064))
