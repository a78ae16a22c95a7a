"""Long self-attention (SelfMask's DINO ViT-S/8 @512x683: 6 heads, dh = 64, T = 5505) with the keys split over S workgroups per
(image, head, query block): us per launch for S = 1 .. 8 at 1 / 2 / 4 / 8 images, split-pair (x3) and plain fp16 operands, next to the
split the engine's model picks (zutis_amd/engine_base.py::long_sequence_key_split).  usage: attn_long_split.py [T]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zutis_amd import ops
from zutis_amd.engine_base import long_sequence_key_split

dev = torch.device("cuda:0")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 5505
H, dh = 6, 64
D = H * dh
for x3 in (True, False):
    for B in (1, 2, 4, 8):
        g = torch.Generator(device=dev).manual_seed(B)
        mk = lambda: torch.randn(B, T, D, device=dev, generator=g)
        def act(t):
            if not x3:
                return t.half()
            hi = t.half()
            return ops.Act(torch.stack([hi, (t - hi.float()).half()]))
        q, k, v = act(mk()), act(mk()), act(mk())
        o = ops.Act(torch.empty((2, B, T, D), device=dev, dtype=torch.float16)) if x3 else torch.empty(B, T, D, device=dev, dtype=torch.float16)
        ktiles = -(-T // (32 if x3 else 64))
        pick = long_sequence_key_split(B * H * -(-T // 128), ktiles, dh, x3, B * T * D)
        row = []
        ref = None
        for S in range(1, 9):
            if S > 1 and (S - 1) * -(-ktiles // S) >= ktiles:
                row.append("   -  ")
                continue
            ws = torch.empty((ops.attention_splitk_workspace_size(B, H, T, dh, S),), dtype=torch.uint8, device=dev) if S > 1 else None
            run = lambda: ops.attention(q, k, v, o, batch=B, heads=H, Tq=T, Tk=T, head_dim=dh, ldq=D, ldk=D, ldv=D, ldo=D, strideQ=T * D, strideK=T * D,
                                        strideV=T * D, strideO=T * D, x3=x3, ksplit=S, workspace=ws)
            for _ in range(2):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 8 * 1e3
            oh = (o.hi if x3 else o).float().clone()
            if ref is None:
                ref = oh
            err = float((oh - ref).abs().max())
            row.append(f"{us:7.1f}{'*' if S == pick else ' '}({err:.0e})")
        tf = 4.0 * B * H * T * T * dh / 1e6
        print(f"{'x3 ' if x3 else 'f16'} B={B}: " + " ".join(row) + f"   [S=1..8 us (max |diff| of the hi plane vs S=1); * = model's pick; {tf / 1e6:.2f} TFLOP]", flush=True)
