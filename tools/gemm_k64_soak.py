"""Developer tool (round 5): race screen of the 64-k-slice ring forms of the one-round GEMM tiles — the split-pair tiles 3064 (64 x 64, four
slots), 6464 (128 x 64, three slots), 7096 / 7128 (128 x 96 / 128 x 128 on the circular ring of 160 LDS pieces) and the plain-fp16 tiles 7032 /
7096 / 7128 — many launches per shape while a second stream keeps the memory system busy, every result compared bit for bit with the
32-k tile's result for the same operands (the K order inside a tile does not depend on the tile).  A refill that overtook a fragment
read, or a counted wait one issue short, would show as rare wrong tiles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zutis_amd import _lib, ops
from zutis_amd.ops import Act
dev = torch.device("cuda:0")
L = _lib.load(raw=True)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
bad = 0
side = torch.cuda.Stream()
junk = torch.randn(64 << 20, device=dev)


def noise():
    with torch.cuda.stream(side):
        junk.mul_(1.0001)


def force(tile):
    _lib.check(L.zh_dev_set_gemm_overrides(0, tile, 0), "zh_dev_set_gemm_overrides")


X3 = [("qkv", 1201, 2304, 768, 96, (7096,)), ("fc", 1201, 3072, 768, 1288, (7128,)), ("out", 1201, 768, 768, 3066, (3064,)), ("n1536", 1201, 1536, 768, 64, (6464,)),
      ("dec", 3200, 768, 768, 96, (7096,)), ("k192", 1201, 768, 192, 64, (3064, 6464, 7096, 7128)), ("ragged", 1111, 1000, 320, 64, (3064, 6464, 7096, 7128))]
for name, M, N, K, base, tiles in X3:
    g = torch.Generator(device=dev).manual_seed(7)
    A32 = torch.randn((M, K), generator=g, device=dev)
    A = Act.empty((M, K), True, dev); ops.cast_f16(A32, A, M, K)
    W = ops.split_weight(torch.randn((N, K), generator=g, device=dev) * 0.03)
    bias = torch.randn((N,), generator=g, device=dev)
    out = Act.empty((M, N), True, dev)
    force(base); ops.gemm_x3(A, W, out, bias=bias); torch.cuda.synchronize()
    ref = out.t.clone()
    for tile in tiles:
        force(tile); n_bad = 0
        for it in range(reps):
            out.t.zero_(); noise()
            ops.gemm_x3(A, W, out, bias=bias)
            torch.cuda.synchronize()
            n_bad += 0 if torch.equal(out.t, ref) else 1
        bad += n_bad
        print(f"x3  {name:7s} {M}x{N}x{K} tile {tile}: {reps} launches, {n_bad} differ from tile {base}", flush=True)
F16 = [("qkv", 1201, 2304, 768, (7096, 7128)), ("proj", 1201, 768, 3072, (7032,)), ("dec", 3200, 768, 768, (7096,)), ("ragged", 1111, 1000, 320, (7032, 7096, 7128))]
for name, M, N, K, tiles in F16:
    g = torch.Generator(device=dev).manual_seed(9)
    A = torch.randn((M, K), generator=g, device=dev).half(); W = (torch.randn((N, K), generator=g, device=dev) * 0.03).half()
    bias = torch.randn((N,), generator=g, device=dev)
    out = torch.empty((M, N), dtype=torch.float32, device=dev)
    force(128); ops.gemm(A, W, out, bias=bias); torch.cuda.synchronize()
    ref = out.clone()
    for tile in tiles:
        force(tile); n_bad = 0
        for it in range(reps):
            out.zero_(); noise()
            ops.gemm(A, W, out, bias=bias)
            torch.cuda.synchronize()
            n_bad += 0 if torch.equal(out, ref) else 1
        bad += n_bad
        print(f"f16 {name:7s} {M}x{N}x{K} tile {tile}: {reps} launches, {n_bad} differ from tile 128", flush=True)
L.zh_dev_set_gemm_overrides(0, 0, 0)
print("SOAK", "FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
