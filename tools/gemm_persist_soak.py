"""Developer tool (round 5): race screen of the persistent big fp16 GEMM tiles — many launches per shape (default grid of 256 workgroups and
forced small grids), every result compared bit for bit with the one-workgroup-per-tile launch of the same operands.  A hand-over race
between a tile's epilogue slabs and the next tile's prologue DMA would show as rare wrong tiles: placed by the barrier, screened here."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zutis_amd import _lib, ops
dev = torch.device("cuda:0")
L = _lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0
for name, M, N, K, f16out, act in [("qkv", 14144, 2304, 768, 1, 0), ("fc", 14144, 3072, 768, 1, 1), ("kv", 56448, 4608, 256, 1, 0), ("c4fc", 8200, 3072, 768, 1, 1),
                                   ("proj192", 14144 * 3, 768, 3072, 0, 0), ("ragged", 9999, 1096, 320, 1, 2), ("ragged32", 7777, 904, 192, 0, 0)]:
    g = torch.Generator(device=dev).manual_seed(11)
    A = torch.randn((M, K), generator=g, device=dev).half(); W = (torch.randn((N, K), generator=g, device=dev) * 0.03).half()
    bias = torch.randn((N,), generator=g, device=dev)
    res = None if f16out else torch.randn((M, N), generator=g, device=dev)
    out = torch.empty((M, N), dtype=torch.float16 if f16out else torch.float32, device=dev)
    def run(p):
        L.zh_dev_set_gemm_persist(p)
        ops.gemm(A, W, out, bias=bias, act=act, residual=res, res_rows=M if res is not None else 0)
    run(0); torch.cuda.synchronize(); ref = out.clone()
    n_bad = 0
    for it in range(reps):
        out.fill_(0)
        run((256, 256, 256, 64, 24)[it % 5])
        torch.cuda.synchronize()
        if not torch.equal(out, ref):
            n_bad += 1
    bad += n_bad
    print(f"{name:9s} {M}x{N}x{K}: {reps} persistent launches, {n_bad} differ from one workgroup per tile", flush=True)
L.zh_dev_set_gemm_persist(256)
print("SOAK", "FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
