"""One shape of the flash-attention kernel, launched a few times: the target of the rocprofv3 --pmc passes
(profiles/r0N_attention_pmc*.json).  usage: attn_pmc_run.py enc|cross|selfmask|selfmask4 [x3]
(selfmask4 = SelfMask's T = 5505 at the pseudo-label batch of 4, with the key split the engine picks for it)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zutis_amd import ops
from zutis_amd.ops import Act
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "enc"
x3 = len(sys.argv) > 2 and sys.argv[2] == "x3"
B, H, dh, Tq, Tk = {"enc": (32, 12, 64, 442, 442), "cross": (32, 8, 96, 100, 1764), "selfmask": (1, 6, 64, 5505, 5505), "selfmask4": (4, 6, 64, 5505, 5505)}[which]
D = H * dh


def mk(T):
    x = torch.randn(B * T, D, device=dev)
    if not x3:
        return x.half()
    a = Act.empty((B * T, D), True, dev)
    ops.cast_f16(x, a, B * T, D)
    return a
q, k, v = mk(Tq), mk(Tk), mk(Tk)
o = Act.empty((B * Tq, D), x3, dev)
S, ws = 1, None
if Tq >= 2048:                                       # the engine's rule for long self-attention (engine_base.long_sequence_key_split)
    from zutis_amd.engine_base import long_sequence_key_split
    S = long_sequence_key_split(B * H * -(-Tq // 128), -(-Tk // (32 if x3 else 64)), dh, x3, B * Tq * D)
    if S > 1:
        ws = torch.empty((ops.attention_splitk_workspace_size(B, H, Tq, dh, S),), dtype=torch.uint8, device=dev)
for _ in range(10):
    ops.attention(q, k, v, o, batch=B, heads=H, Tq=Tq, Tk=Tk, head_dim=dh, ldq=D, ldk=D, ldv=D, ldo=D, strideQ=Tq * D, strideK=Tk * D, strideV=Tk * D,
                  strideO=Tq * D, x3=x3, ksplit=S, workspace=ws)
torch.cuda.synchronize()
