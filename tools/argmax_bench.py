import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zutis_amd import ops
dev = torch.device("cuda:0")
for (B, n, h, H) in ((32, 81, 42, 336), (16, 920, 64, 518)):
    lo = torch.randn(B, n, h, h, device=dev); lab = torch.empty(B, H, H, dtype=torch.int64, device=dev)
    for _ in range(3): ops.upsample_argmax(lo, lab, B, n, h, h, H, H)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): ops.upsample_argmax(lo, lab, B, n, h, h, H, H)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    byts = lo.numel() * 4 + lab.numel() * 8
    print(f"B={B} n={n} {h}->{H}: {dt*1e6:.1f} us  ({byts/1e6:.0f} MB compulsory -> {byts/dt/1e12:.2f} TB/s; {B*H*H*n/dt/1e12:.2f} T pixel-classes/s)")
