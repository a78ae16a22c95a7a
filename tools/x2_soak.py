"""Developer soak of the two-product (SPLIT = 2) three-slot loop: every output of N repeats must be bitwise the x3 kernel's, while a
second stream runs other GEMMs to perturb the timing (a mis-counted vmcnt / missing barrier shows up as rare wrong tiles).
   gpurun -- python tools/x2_soak.py [repeats]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zutis_amd import ops
from zutis_amd.ops import Act
dev = torch.device("cuda:0")
N_REP = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = torch.Generator().manual_seed(0)
side = torch.cuda.Stream()
bad = 0
for (M, N, K, kind) in [(14144, 2304, 768, "split"), (14144, 768, 3072, "f32"), (14144, 768, 768, "f32"), (36928, 3072, 1024, "split"),
                        (36928, 1024, 4096, "f32"), (4500, 4608, 256, "split"), (4100, 776, 64, "f32"), (65755, 1024, 128, "f32")]:
    A32 = torch.randn((M, K), generator=g).to(dev)
    W32 = (torch.randn((N, K), generator=g) * 0.05).half().float().to(dev)
    A = Act.empty((M, K), True, dev); ops.cast_f16(A32, A, M, K)
    W2, W3 = ops.split_weight(W32), ops.split_weight(W32, allow_x2=False)
    res = torch.randn((M, N), generator=g).to(dev) if kind == "f32" else None
    mk = (lambda: torch.empty((M, N), device=dev)) if kind == "f32" else (lambda: Act.empty((M, N), True, dev))
    ref = mk(); ops.gemm_x3(A, W3, ref, residual=res)
    reft = ref if kind == "f32" else ref.t
    # perturbing work on the side stream
    Bs = Act.empty((3000, 768), True, dev); ops.cast_f16(torch.randn((3000, 768), device=dev), Bs, 3000, 768)
    Ws = ops.split_weight(torch.randn((768, 768), device=dev) * 0.05)
    os_ = torch.empty((3000, 768), device=dev)
    torch.cuda.synchronize()
    nbad = 0
    out = mk()
    for i in range(N_REP):
        with torch.cuda.stream(side):
            for _ in range(1 + i % 3):
                ops.gemm_x3(Bs, Ws, os_)
        ops.gemm_x3(A, W2, out, residual=res)
        o = out if kind == "f32" else out.t
        if not torch.equal(o, reft):
            nbad += 1
    torch.cuda.synchronize()
    bad += nbad
    print(f"{M}x{N}x{K} {kind}: {N_REP} repeats, {nbad} differing from the x3 result", flush=True)
print("SOAK", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
