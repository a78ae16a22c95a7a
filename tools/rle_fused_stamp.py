"""Developer: phase stamps of zh_mask_rle_fused_kept (a variant library built with -DZH_FRLE_STAMP:
   bash tools/build_variant_lib.sh tools/_abl/libzutis_frle.so -DZH_FRLE_STAMP instance.hip) on the config-3 fixture's kept masks.
   gpurun -- env ZUTIS_HIP_LIB=$PWD/tools/_abl/libzutis_frle.so python tools/rle_fused_stamp.py [bytes]      (bytes: pack from the u8 masks)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from zutis_amd import ops
dev = torch.device("cuda:0")
g = np.load(os.path.join(ROOT, "tests", "golden", "c3_vitb16.npz"))
m = np.unpackbits(g["480x640_masks"], axis=2)                       # the reference's 17 kept masks, [17, 480, 640]
n, H, W = m.shape
Q = 100
masks = torch.zeros((1, Q, H, W), dtype=torch.uint8, device=dev)
masks[0, :n] = torch.from_numpy(m).to(dev)
idx = torch.arange(Q, dtype=torch.int32, device=dev).view(1, Q)
cnt = torch.tensor([n], dtype=torch.int32, device=dev)
cap = 1 << 20
out = torch.zeros((cap,), dtype=torch.uint8, device=dev)
cursor = torch.zeros((1,), dtype=torch.int32, device=dev)
info = torch.zeros((Q, 8), dtype=torch.int32, device=dev)
bits = torch.empty((1, Q, (H * W + 63) // 64), dtype=torch.int64, device=dev) if "bytes" not in sys.argv[1:] else None
if bits is not None:
    ops.mask_iou_counts(masks[0], Q, H * W, torch.empty((Q, Q), dtype=torch.int32, device=dev), torch.empty((Q, Q), dtype=torch.int32, device=dev),
                        workspace=bits[0])
for _ in range(3):
    cursor.zero_(); ops.mask_rle_fused_kept(masks, idx, cnt, 8192, out, cursor, info, bits=bits)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.mask_rle_fused_kept(masks, idx, cnt, 8192, out, cursor, info, bits=bits)
e1.record(); torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch ({n} kept masks of {H}x{W}, transitions {info[:n, 7].tolist()})")
st = out.cpu().numpy()
names = ["entry -> bits in LDS", "pass 1 (count per chunk, column)", "column prefix + scan", "pass 2 (write the list)", "string into LDS + cursor", "string copied out"]
rows = []
for mi in range(n):
    t = st[cap - 64 * (mi + 1):cap - 64 * (mi + 1) + 56].view(np.int64).astype(np.float64) / 100.0
    rows.append(np.diff(t))
rows = np.array(rows)
if rows.size and rows.max() > 0:
    for i, nm in enumerate(names):
        print(f"   {nm:34s} {rows[:, i].mean():6.2f} us (max {rows[:, i].max():.2f})")
else:
    print("   (no stamps: the library was not built with -DZH_FRLE_STAMP)")
