"""Developer tool (round 5): the split-pair (x3) dh = 64 flash attention with one (ZH_ATTN_QT=1, product) or two (ZH_ATTN_QT=2) 32-query
tiles per wave, on the model's encoder shapes.  The switch is read once per process: run this script under both values alternately on
ONE box (bash tools/attn_qt_ab.sh); it prints us per launch and a checksum of the output (the two forms must agree bit for bit: the
same arithmetic per query)."""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zutis_amd import ops
from zutis_amd.ops import Act
dev = torch.device("cuda:0")
out = []
for name, B, H, dh, T in [("enc", 32, 12, 64, 442), ("c4enc", 8, 12, 64, 1025), ("c5enc", 256, 16, 64, 577), ("selfmask", 4, 6, 64, 5505), ("b1enc", 1, 12, 64, 1201)]:
    D = H * dh
    g = torch.Generator(device=dev).manual_seed(3)
    def pair(t):
        a = Act.empty(tuple(t.shape), True, dev)
        ops.cast_f16(t.reshape(-1, t.shape[-1]).contiguous(), a, t.numel() // t.shape[-1], t.shape[-1])
        return a
    q, k, v = (pair(torch.randn((B * T, D), generator=g, device=dev)) for _ in range(3))
    o = Act.empty((B * T, D), True, dev)
    run = lambda: ops.attention(q, k, v, o, batch=B, heads=H, Tq=T, Tk=T, head_dim=dh, ldq=D, ldk=D, ldv=D, ldo=D, strideQ=T * D, strideK=T * D,
                                strideV=T * D, strideO=T * D, x3=True)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30 if B * T < 100000 else 8
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    h = hashlib.md5(o.t.cpu().numpy().tobytes()).hexdigest()[:8]
    out.append(f"{name}: {us:7.1f} us ({4.0 * B * H * T * T * dh / us / 1e6:4.0f} TF) {h}")
print(f"QT={os.environ.get('ZH_ATTN_QT', '1')}  " + "  ".join(out))
