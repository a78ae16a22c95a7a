"""One shape of the flash-attention kernel, launched a few times: the target of the rocprofv3 --pmc passes
(profiles/r02_attention_pmc.md)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zutis_amd import ops
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "enc"
B, H, dh, Tq, Tk = {"enc": (32, 12, 64, 442, 442), "cross": (32, 8, 96, 100, 1764), "selfmask": (1, 6, 64, 5505, 5505)}[which]
D = H * dh
q = torch.randn(B, Tq, D, device=dev).half(); k = torch.randn(B, Tk, D, device=dev).half(); v = torch.randn(B, Tk, D, device=dev).half()
o = torch.empty(B, Tq, D, device=dev, dtype=torch.float16)
for _ in range(10):
    ops.attention(q, k, v, o, batch=B, heads=H, Tq=Tq, Tk=Tk, head_dim=dh, ldq=D, ldk=D, ldv=D, ldo=D, strideQ=Tq * D, strideK=Tk * D, strideV=Tk * D, strideO=Tq * D)
torch.cuda.synchronize()
