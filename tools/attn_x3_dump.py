import sys, os, math, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zutis_amd import ops
from zutis_amd.ops import Act
dev = torch.device("cuda:0")
f16 = torch.float16
def split(x):
    hi = x.to(f16); lo = (x - hi.float()).to(f16)
    return Act(torch.stack([hi, lo]).contiguous().to(dev))
dh, heads, Tq, Tk, B = 64, 1, 442, 442, 2
g = torch.Generator().manual_seed(1)
D = heads * dh
q = torch.randn((B * Tq, D), generator=g) * 2.5; k = torch.randn((B * Tk, D), generator=g) * 2.5; v = torch.randn((B * Tk, D), generator=g)
kw = dict(batch=B, heads=heads, Tq=Tq, Tk=Tk, head_dim=dh, ldq=D, ldk=D, ldv=D, ldo=D, strideQ=Tq * D, strideK=Tk * D, strideV=Tk * D, strideO=Tq * D)
Q, K, V = split(q), split(k), split(v)
O3 = Act.empty((B * Tq, D), True, dev); ops.attention(Q, K, V, O3, x3=True, **kw)
V0 = Act(torch.stack([V.t[0], torch.zeros_like(V.t[1])]).contiguous())
O3v0 = Act.empty((B * Tq, D), True, dev); ops.attention(Q, K, V0, O3v0, x3=True, **kw)
np.savez_compressed("gpurun_out/attn_dump.npz", hi=O3.t[0].cpu().numpy(), lo=O3.t[1].cpu().numpy(), hi_v0=O3v0.t[0].cpu().numpy(), lo_v0=O3v0.t[1].cpu().numpy())
