"""Developer: where the cycles of one flash-attention key tile go, from in-kernel s_memtime stamps (a diagnostic build of
attention.hip with -DZH_ATTN_STAMP; the product kernel has no stamps).  Segments per tile, per wave: 0 tile-load issue, 1 K
fragment reads + K.Q^T MFMA issue, 2 wait for S + row max, 3 exp / split / pack (+ rescale), 4 V reads + P.V MFMA issue,
5 tile stores (incl. the wait for the global loads), 6 barrier.  The in-kernel clock is d(s_memtime) / d(s_memrealtime) x 100 MHz,
read after >= 2 s of back-to-back launches (MI355X_MICROARCH.md, DVFS give-back item 6).
  python tools/attn_stamp.py --build      (here, hipcc)        gpurun -- python tools/attn_stamp.py [--x3] [shape ...]"""
import ctypes as C, os, subprocess, sys, time
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "_abl", "libattn_stamp.so")
if "--build" in sys.argv:
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-DZH_ATTN_STAMP"] +
                          [a for a in sys.argv[1:] if a.startswith("-D")] +
                          [os.path.join(ROOT, "zutis_amd/csrc/attention.hip"), os.path.join(ROOT, "zutis_amd/csrc/capi.hip"), "-o", LIB])
    sys.exit(0)
import numpy as np, torch
dev = torch.device("cuda:0")
vp, l, i, f = C.c_void_p, C.c_long, C.c_int, C.c_float
L = C.CDLL(LIB)
L.zh_attention_f16.restype = i
L.zh_attention_f16.argtypes = [vp, l, l, vp, l, l, vp, l, l, vp, l, l, i, i, i, i, i, f, l, l, l, l, vp]
L.zh_attn_set_stamp.argtypes = [vp]
SHAPES = {"enc": (32, 12, 64, 442, 442), "cross": (32, 8, 96, 100, 1764), "c4enc": (8, 12, 64, 1025, 1025), "c5enc": (256, 16, 64, 577, 577),
          "selfmask": (1, 6, 64, 5505, 5505)}
X3 = "--x3" in sys.argv
names = [a for a in sys.argv[1:] if a in SHAPES] or ["enc", "cross"]
SEG = ["load issue", "K reads + S issue", "S wait + max", "exp/split/pack", "V reads + PV issue", "tile store (+vmcnt)", "barrier"]
for name in names:
    B, H, dh, Tq, Tk = SHAPES[name]
    D = H * dh; P = 2 if X3 else 1
    q = torch.randn(P, B, Tq, D, device=dev).half(); k = torch.randn(P, B, Tk, D, device=dev).half(); v = torch.randn(P, B, Tk, D, device=dev).half()
    if X3:
        q[1] *= 2 ** -11; k[1] *= 2 ** -11; v[1] *= 2 ** -11
    o = torch.empty(P, B, Tq, D, device=dev, dtype=torch.float16)
    pq, pk, po = (B * Tq * D, B * Tk * D, B * Tq * D) if X3 else (0, 0, 0)
    nqb = (Tq + 127) // 128
    nwg = ((B * H + 7) // 8) * 8 * nqb
    stamp = torch.zeros(nwg * 4 * 16, dtype=torch.int64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    run = lambda: L.zh_attention_f16(q.data_ptr(), D, Tq * D, k.data_ptr(), D, Tk * D, v.data_ptr(), D, Tk * D, o.data_ptr(), D, Tq * D,
                                     B, H, Tq, Tk, dh, dh ** -0.5, pq, pk, pk, po, s)
    L.zh_attn_set_stamp(None)
    assert run() == 0
    torch.cuda.synchronize()
    t0 = time.time()
    while time.time() - t0 < 2.0:                      # the clock the chip settles at under this kernel
        for _ in range(50): run()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us_plain = e0.elapsed_time(e1) / 20 * 1e3
    L.zh_attn_set_stamp(stamp.data_ptr())
    for _ in range(20): run()
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    st = stamp.cpu().numpy().reshape(nwg, 4, 16).astype(np.float64)
    act = st[:, :, 11] > 0                             # waves with queries
    used = st[:, :, 10] > 0
    a = st[act & used]
    tiles = a[:, 10].mean()
    clk = (a[:, 8] / a[:, 9]).mean() * 100e6 / 1e9
    per = a[:, :7].sum(0) / a[:, 10].sum()
    print(f"{name} {'x3' if X3 else 'f16'}: {us_plain:.1f} us (stamp stores off) / {us:.1f} us (on); in-kernel clock {clk:.2f} GHz; wave lifetime "
          f"{a[:, 8].mean():.0f} cycles for {tiles:.1f} tiles = {a[:, 8].mean() / tiles:.0f} per tile; {int(act.sum())} active waves of {int(used.sum())}")
    for nm, c in zip(SEG, per):
        print(f"    {nm:22s} {c:7.0f} cycles / tile  ({100 * c / per.sum():4.1f} %)")
    u = st[used]
    t0 = u[:, 12].min()
    start, loop_end, end = (u[:, 12] - t0) / 100.0, (u[:, 13] - t0) / 100.0, (u[:, 14] - t0) / 100.0          # us since the first wave's entry
    print(f"    prologue {u[:, 7].mean():.0f} cycles; timeline (us since the first entry): last entry {start.max():.1f}, entries after 5 us: "
          f"{(start > 5).mean() * 100:.0f} %, median loop end {np.median(loop_end):.1f}, last loop end {loop_end.max():.1f}, last store done {end.max():.1f}")
    hs, _ = np.histogram(start, bins=10, range=(0, end.max())); he, _ = np.histogram(end, bins=10, range=(0, end.max()))
    print(f"    entries per tenth of the span {hs.tolist()}  exits {he.tolist()}")
    idle = st[used & ~act]
    if len(idle):
        print(f"    (query-less waves: lifetime {idle[:, 8].mean():.0f} cycles)")
