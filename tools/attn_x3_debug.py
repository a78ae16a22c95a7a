import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zutis_amd import ops
from zutis_amd.ops import Act
dev = torch.device("cuda:0")
f16 = torch.float16
def split(x):
    hi = x.to(f16); lo = (x - hi.float()).to(f16)
    return Act(torch.stack([hi, lo]).contiguous().to(dev))
def run(dh, heads, Tq, Tk, causal, qs=2.5, B=2):
    g = torch.Generator().manual_seed(1)
    D = heads * dh
    q = torch.randn((B * Tq, D), generator=g) * qs; k = torch.randn((B * Tk, D), generator=g) * qs; v = torch.randn((B * Tk, D), generator=g)
    qd, kd, vd = (t.view(B, -1, heads, dh).transpose(1, 2).double() for t in (q, k, v))
    s = qd @ kd.transpose(-1, -2) / math.sqrt(dh)
    if causal: s = s + torch.full((Tq, Tk), float("-inf"), dtype=torch.float64).triu_(1)
    p = torch.softmax(s, -1)
    ref = (p @ vd).transpose(1, 2).reshape(B * Tq, D)
    kw = dict(batch=B, heads=heads, Tq=Tq, Tk=Tk, head_dim=dh, ldq=D, ldk=D, ldv=D, ldo=D, strideQ=Tq * D, strideK=Tk * D, strideV=Tk * D, strideO=Tq * D, causal=causal)
    Q, K, V = split(q), split(k), split(v)
    O3 = Act.empty((B * Tq, D), True, dev); ops.attention(Q, K, V, O3, x3=True, **kw)
    o3 = (O3.t[0].float() + O3.t[1].float()).cpu().double()
    O1 = Act.empty((B * Tq, D), False, dev); ops.attention(Q, K, V, O1, x3=False, **kw)
    e3 = (o3 - ref).abs(); e1 = (O1.hi.float().cpu().double() - ref).abs()
    i = int(e3.argmax()); r, c = i // D, i % D
    print(f"dh{dh} h{heads} Tq{Tq} Tk{Tk} causal{causal} qs{qs}: e3 {e3.max():.2e} (row {r} = img {r//Tq} q {r%Tq}, col {c} = head {c//dh} d {c%dh}) e1 {e1.max():.2e}  pmax@worst {float(p[r//Tq, c//dh, r%Tq].max()):.3f}  mean e3 {e3.mean():.2e}")
    # error per query position / head
    e3v = e3.view(B, Tq, heads, dh).amax(-1)
    print("   worst per head:", [f"{float(e3v[:, :, h].max()):.1e}" for h in range(heads)], " per image:", [f"{float(e3v[b].max()):.1e}" for b in range(B)],
          " by q-block of 32:", [f"{float(e3v[:, a:a+32].max()):.0e}" for a in range(0, Tq, 32)][:16])
for cfg in [(64, 3, 442, 442, False), (96, 2, 100, 1764, False), (64, 2, 77, 77, True), (64, 1, 130, 700, False), (64, 1, 442, 442, False), (64, 3, 130, 700, False), (64, 3, 442, 442, False, 1.0)]:
    run(*cfg)
