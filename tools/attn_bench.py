"""Attention micro-benchmark over the model's shapes, 2- vs 4-wave blocks (developer tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zutis_amd import ops
dev = torch.device("cuda:0")
for name, B, H, dh, Tq, Tk in [("enc", 32, 12, 64, 442, 442), ("cross", 32, 8, 96, 100, 1764), ("self", 32, 8, 96, 100, 100), ("c4enc", 8, 12, 64, 1025, 1025),
                              ("selfmask", 1, 6, 64, 5505, 5505)]:
    D = H * dh
    q = torch.randn(B, Tq, D, device=dev).half(); k = torch.randn(B, Tk, D, device=dev).half(); v = torch.randn(B, Tk, D, device=dev).half()
    o = torch.empty(B, Tq, D, device=dev, dtype=torch.float16)
    res = {}
    for wv in ("auto", "2", "4"):
        if wv == "auto": os.environ.pop("ZH_ATTN_WAVES", None)
        else: os.environ["ZH_ATTN_WAVES"] = wv
        def run(n):
            for _ in range(n):
                ops.attention(q, k, v, o, batch=B, heads=H, Tq=Tq, Tk=Tk, head_dim=dh, ldq=D, ldk=D, ldv=D, ldo=D, strideQ=Tq * D, strideK=Tk * D, strideV=Tk * D, strideO=Tq * D)
        run(3); ts = []
        for r in range(3):
            torch.cuda.synchronize(); t = time.perf_counter(); run(20); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) / 20 * 1e6)
        res[wv] = min(ts)
    fl = 4.0 * B * H * Tq * Tk * dh
    print(f"{name:8s}", " ".join(f"{k}:{v:8.1f}us ({fl / v / 1e6:6.1f} TF)" for k, v in res.items()))
