"""Fused bilinear upsample + argmax (networks/zutis.py:366-372): us per launch of the product kernel next to the round-2 kernel
(ZH_UPSAMPLE_ARGMAX_PK=0 in a child process), labels compared bit for bit, random logits and spatially smooth ones.
usage: argmax_bench.py [child]"""
import os
import subprocess
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zutis_amd import ops

dev = torch.device("cuda:0")
CASES = ((32, 81, 42, 336), (8, 920, 74, 518), (16, 920, 64, 518), (1, 81, 60, 480), (4, 21, 42, 336))


def logits(B, n, h, smooth):
    g = torch.Generator(device=dev).manual_seed(n + h)
    lo = torch.randn(B, n, h, h, device=dev, generator=g)
    if smooth:       # neighbouring pixels share their leading classes, as on images
        lo = torch.nn.functional.avg_pool2d(lo, 5, stride=1, padding=2) * 3
    return lo.contiguous()


def run():
    out = {}
    for (B, n, h, H) in CASES:
        for smooth in (False, True):
            lo = logits(B, n, h, smooth)
            lab = torch.empty(B, H, H, dtype=torch.int64, device=dev)
            for _ in range(3):
                ops.upsample_argmax(lo, lab, B, n, h, h, H, H)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(10):
                ops.upsample_argmax(lo, lab, B, n, h, h, H, H)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
            out[(B, n, h, H, smooth)] = (dt, int(lab.sum().item()), int((lab * torch.arange(lab.numel(), device=dev).view_as(lab) % 1000003).sum().item()))
    return out


if len(sys.argv) > 1 and sys.argv[1] == "child":
    for k, v in run().items():
        print("R", *k, *v)
    sys.exit(0)
new = run()
env = {**os.environ, "ZH_UPSAMPLE_ARGMAX_PK": "0"}
r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
old = {}
for ln in r.stdout.splitlines():
    if ln.startswith("R "):
        f = ln.split()
        old[(int(f[1]), int(f[2]), int(f[3]), int(f[4]), f[5] == "True")] = (float(f[6]), int(f[7]), int(f[8]))
for k, (dt, s1, s2) in new.items():
    B, n, h, H, smooth = k
    o = old.get(k)
    byts = B * n * h * h * 4 + B * H * H * 8
    print(f"B={B} n={n} {h}->{H} {'smooth' if smooth else 'random'}: {dt * 1e6:8.1f} us (round-2 kernel {o[0] * 1e6:8.1f} us, x{o[0] / dt:.2f}); "
          f"labels identical: {o[1:] == (s1, s2)}; {byts / 1e6:.0f} MB compulsory -> {byts / dt / 1e12:.2f} TB/s; {B * H * H * n / dt / 1e12:.2f} T pixel-classes/s")
