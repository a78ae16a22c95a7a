import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zutis_amd import ops
dev = torch.device("cuda:0")
B, h, w = 32, 21, 21
for C, f16out in ((2048, True), (512, False), (768, True)):
    x = torch.randn(B, h, w, C, device=dev)
    bias = torch.randn(C, device=dev)
    o16 = torch.empty(B * 4 * h * w, C, dtype=torch.float16, device=dev)
    o32 = torch.empty(B * 4 * h * w, C, dtype=torch.float32, device=dev)
    def run():
        if f16out: ops.upsample2x_cl(x, B, h, w, C, out_f16=o16, bias=bias, act=ops.ACT_RELU)
        else: ops.upsample2x_cl(x, B, h, w, C, out_f32=o32)
    for _ in range(3): run()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    byts = x.numel() * 4 + (o16.numel() * 2 if f16out else o32.numel() * 4)
    print(f"C={C} f16out={f16out}: {dt*1e6:.1f} us, compulsory {byts/1e6:.0f} MB -> {byts/dt/1e12:.2f} TB/s")
