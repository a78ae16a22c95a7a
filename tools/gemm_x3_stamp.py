"""Developer: per-block phase timestamps of the f16x3 GEMM's two-slot tiles (a variant library built with -DZH_GEMM_PROBE:
   bash tools/build_variant_lib.sh tools/_abl/libzutis_probe.so -DZH_GEMM_PROBE gemm_x3.hip).  Phases: entry -> first slice issued
(+ pos tables) -> K loop done -> output stores drained; 100-MHz wall clock and shader cycles.
   gpurun -- env ZUTIS_HIP_LIB=$PWD/tools/_abl/libzutis_probe.so python tools/gemm_x3_stamp.py [x2]
`f16` (round 5): the PLAIN fp16 kernel on the encoder and ViT-L/14 shapes (variant library built from gemm.hip instead:
   bash tools/build_variant_lib.sh tools/_abl/libzutis_probe16.so -DZH_GEMM_PROBE gemm.hip), one workgroup per tile (persist = 0) so that
every tile leaves its own stamps; prints the MFMA pipe's busy share of the K loop at the clock the stamps read."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, numpy as np, torch
from zutis_amd import ops, _lib
from zutis_amd.ops import Act
dev = torch.device("cuda:0")
raw = _lib.load(raw=True)
F16 = "f16" in sys.argv[1:]
set_probe = raw.zh_gemm_set_probe if F16 else raw.zh_gemm_x3_set_probe
set_probe.argtypes = [C.c_void_p]
shapes = [(56448, 4608, 256, "kv-all", "split"), (14144, 2304, 768, "qkv", "split"), (14144, 3072, 768, "fc", "split"), (14144, 768, 768, "out", "f32"),
          (14144, 768, 3072, "proj", "f32")]
X2 = "x2" in sys.argv[1:]          # fp16-valued weights: the two-product kernel (SPLIT = 2), plus the ViT-L/14 shapes of config 5
if X2:
    shapes = shapes[1:] + [(147712, 3072, 1024, "L qkv", "split"), (147712, 1024, 1024, "L out", "f32"), (147712, 4096, 1024, "L fc", "split"),
                           (147712, 1024, 4096, "L proj", "f32")]
if F16:
    shapes = [(14144, 2304, 768, "qkv", "f16"), (14144, 3072, 768, "fc", "f16"), (56448, 4608, 256, "kv-all", "f16"), (14144, 768, 768, "out", "f32"),
              (14144, 768, 3072, "proj", "f32"), (147712, 3072, 1024, "L qkv", "f16"), (147712, 4096, 1024, "L fc", "f16"), (147712, 1024, 4096, "L proj", "f32")]
    _lib.check(raw.zh_dev_set_gemm_persist(0))
B1 = "b1" in sys.argv[1:]          # batch-1 evaluation shapes (config 3: T = 1201 tokens), forced tile codes: b1 [tile ...]
if B1:
    shapes = [(1201, 2304, 768, "qkv", "split"), (1201, 3072, 768, "fc", "split"), (1201, 768, 768, "out", "f32"), (1201, 768, 3072, "proj", "f32"),
              (100, 768, 768, "dec", "f32")]
    tiles = [int(a) for a in sys.argv[1:] if a.isdigit()] or [0]
    shapes = [s + (t,) for s in shapes for t in tiles]
for sh in shapes:
    M, N, K, name, kind = sh[:5]
    if len(sh) > 5:
        _lib.check(raw.zh_dev_set_gemm_overrides(0, sh[5], 0)); name = f"{name}/t{sh[5]}"
    A32 = torch.randn(M, K, device=dev); W32 = torch.randn(N, K, device=dev) * 0.03
    if X2:
        W32 = W32.half().float()
    if F16:
        A, W = A32.half(), W32.half()
        out = torch.empty(M, N, device=dev) if kind == "f32" else torch.empty(M, N, device=dev, dtype=torch.float16)
        res = out if kind == "f32" else None
        run = lambda: ops.gemm(A, W, out, residual=res, act=ops.ACT_QUICKGELU if "fc" in name else 0)
    else:
        A = Act.empty((M, K), True, dev); ops.cast_f16(A32, A, M, K)
        W = ops.split_weight(W32)
        out = torch.empty(M, N, device=dev) if kind == "f32" else Act.empty((M, N), True, dev)
        res = out if kind == "f32" else None
        run = lambda: ops.gemm_x3(A, W, out, residual=res)
    probe = torch.zeros(16384 * 8, dtype=torch.int64, device=dev)
    set_probe(None)
    for _ in range(10): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    set_probe(probe.data_ptr())
    if B1:                                  # the probed launch reads a weight that was never touched: from HBM, as in the model
        W = ops.split_weight(torch.randn(N, K, device=dev) * 0.03)
        junk = torch.empty(300 << 20, dtype=torch.uint8, device=dev); junk.fill_(1); torch.cuda.synchronize()
    run(); torch.cuda.synchronize()
    set_probe(None)
    r = probe.cpu().numpy().reshape(-1, 8)
    r = r[r[:, 0] > 0]
    t = r[:, :4].astype(np.float64) / 100.0
    t -= t[:, 0].min()
    clk = ((r[:, 6] - r[:, 5]) / np.maximum(r[:, 2] - r[:, 1], 1)).mean() * 0.1
    pro, loop, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    print(f"{name:7s} {M}x{N}x{K} {kind}: {us:.1f} us; {len(r)} blocks, span {t[:, 3].max():.1f} us, clock in the K loop {clk:.2f} GHz")
    if F16:       # 8 waves on 4 SIMDs, 16x16x32 MFMAs of 16 cycles: a 32-k slice is (BM / 16) (BN / 16) / 4 x 16 cycles of MFMA work per SIMD
        bn = 192 if (N == 768 and M == 14144) else 256
        cyc = (256 // 16) * (bn // 16) / 4 * 16 * (K // 32)
        print(f"    MFMA pipe busy in the K loop: {cyc / (loop.mean() * clk * 1e3):.2f} ({cyc:.0f} cycles of MFMA per SIMD in {loop.mean():.2f} us at {clk:.2f} GHz)")
    first = t[:, 0] < 2.0
    for lab, sel in (("first-round blocks", first), ("later blocks", ~first)):
        if sel.sum():
            print(f"    {lab:18s} n={int(sel.sum()):5d}  prologue {pro[sel].mean():6.2f}  K loop {loop[sel].mean():6.2f} (min {loop[sel].min():.2f}, max {loop[sel].max():.2f})  "
                  f"epilogue {epi[sel].mean():6.2f} (max {epi[sel].max():.2f})  block {(t[sel, 3] - t[sel, 0]).mean():6.2f} us")
