"""Encoder self-attention at batch 1 (T tokens, 12 heads, dh 64, split pairs) against the key split, and the decoder's attentions:
us per launch from a hipGraph replay.  usage: attn_small_bench.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zutis_amd import ops
from zutis_amd.ops import Act
from gemm_small_bench import t
dev = torch.device("cuda:0")


def run(T, Tk, heads, dh, splits, self_attn=True):
    D = heads * dh
    if self_attn:
        qkv = Act.empty((T, 3 * D), True, dev); ops.cast_f16(torch.randn(T, 3 * D, device=dev), qkv, T, 3 * D)
        q, k, v = qkv, qkv.view(qkv.hi[:, D:]), qkv.view(qkv.hi[:, 2 * D:])
        ld = dict(ldq=3 * D, ldk=3 * D, ldv=3 * D, strideQ=T * 3 * D, strideK=T * 3 * D, strideV=T * 3 * D)
    else:
        q = Act.empty((T, D), True, dev); ops.cast_f16(torch.randn(T, D, device=dev), q, T, D)
        k = Act.empty((Tk, D), True, dev); ops.cast_f16(torch.randn(Tk, D, device=dev), k, Tk, D)
        v = Act.empty((Tk, D), True, dev); ops.cast_f16(torch.randn(Tk, D, device=dev), v, Tk, D)
        ld = dict(ldq=D, ldk=D, ldv=D, strideQ=T * D, strideK=Tk * D, strideV=Tk * D)
    o = Act.empty((T, D), True, dev)
    line = f"T={T} Tk={Tk} heads={heads} dh={dh}:"
    ref = None
    for S in splits:
        ws = torch.empty(max(16, ops.attention_splitk_workspace_size(1, heads, T, dh, S)), dtype=torch.uint8, device=dev) if S > 1 else None
        try:
            fn = lambda i: ops.attention(q, k, v, o, batch=1, heads=heads, Tq=T, Tk=Tk, head_dim=dh, ldo=D, strideO=T * D, x3=True, ksplit=S, workspace=ws, **ld)
            fn(0); torch.cuda.synchronize()
            got = o.t[0].float() + o.t[1].float()
            if ref is None: ref = got.clone()
            line += f" | S{S}: {t(fn):5.1f}us d{float((got - ref).abs().max()):.0e}"
        except Exception as e:
            line += f" | S{S}: ERR {str(e)[:50]}"
    print(line, flush=True)


run(1201, 1201, 12, 64, (1, 2, 3, 4))
run(442, 442, 12, 64, (1, 2, 3, 4, 7))
run(1025, 1025, 12, 64, (1, 2, 3, 4))
run(100, 100, 8, 96, (1,))
run(100, 4800, 8, 96, (8, 12, 16, 24, 30), self_attn=False)
run(100, 1764, 8, 96, (8, 12, 16, 24), self_attn=False)
