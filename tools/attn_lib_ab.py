"""Developer tool (round 6): the split-pair (x3) flash attention of whichever library ZUTIS_HIP_LIB names, on the model's shapes —
us per launch, and the max |diff| of both output planes' SUM against an fp64 softmax reference on a sample of rows (fp32-class or not).
Run under two libraries alternately on ONE box (bash tools/attn_lib_ab.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zutis_amd import ops
from zutis_amd.ops import Act
dev = torch.device("cuda:0")
out = []
SHAPES = [("enc", 32, 12, 64, 442, 442), ("c4enc", 8, 12, 64, 1025, 1025), ("c5enc", 256, 16, 64, 577, 577), ("selfmask4", 4, 6, 64, 5505, 5505),
          ("b1enc", 1, 12, 64, 1201, 1201), ("decself", 32, 8, 96, 100, 100), ("cross", 32, 8, 96, 100, 1764)]
for name, B, H, dh, Tq, Tk in SHAPES:
    D = H * dh
    g = torch.Generator(device=dev).manual_seed(3)
    def pair(t):
        a = Act.empty(tuple(t.shape), True, dev)
        ops.cast_f16(t.reshape(-1, t.shape[-1]).contiguous(), a, t.numel() // t.shape[-1], t.shape[-1])
        return a
    qf, kf, vf = torch.randn((B * Tq, D), generator=g, device=dev) * 1.5, torch.randn((B * Tk, D), generator=g, device=dev) * 1.5, torch.randn((B * Tk, D), generator=g, device=dev)
    q, k, v = pair(qf), pair(kf), pair(vf)
    o = Act.empty((B * Tq, D), True, dev)
    run = lambda: ops.attention(q, k, v, o, batch=B, heads=H, Tq=Tq, Tk=Tk, head_dim=dh, ldq=D, ldk=D, ldv=D, ldo=D, strideQ=Tq * D, strideK=Tk * D,
                                strideV=Tk * D, strideO=Tq * D, x3=True)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30 if B * Tq * Tk < 3e7 else 8
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    # fp64 reference: image 0, head 0 and the last head, the operands as the kernel sees them (hi + lo)
    got = (o.t[0].float() + o.t[1].float()).view(B, Tq, D)
    err = 0.0
    for h in (0, H - 1):
        sl = slice(h * dh, (h + 1) * dh)
        Q = (q.t[0].double() + q.t[1].double()).view(B, Tq, D)[0, :, sl]
        K = (k.t[0].double() + k.t[1].double()).view(B, Tk, D)[0, :, sl]
        V = (v.t[0].double() + v.t[1].double()).view(B, Tk, D)[0, :, sl]
        ref = torch.softmax(Q @ K.T / dh ** 0.5, dim=1) @ V
        err = max(err, float((got[0, :, sl].double() - ref).abs().max()))
    out.append(f"{name}: {us:7.1f} us ({4.0 * B * H * Tq * Tk * dh / us / 1e6:4.0f} TF) err {err:.1e}")
print(f"{os.path.basename(os.environ.get('ZUTIS_HIP_LIB', 'product'))[:18]:18s} " + "  ".join(out), flush=True)
