"""Where an attention key tile's time goes: the kernel rebuilt with pieces removed (-DZH_ATTN_ABL=mask: 1 no exp, 2 no P.V,
4 no K.Q^T, 8 no tile traffic / barriers, 16 no barriers, 32 barriers only, 64 no LDS stores), timed on the encoder and cross-attention shapes.
CAVEAT: a variant that stops writing the LDS tiles computes on constant data and clocks higher (power) — the
no-traffic columns overstate what the traffic costs.  `--build` compiles the variants
(hipcc, no GPU needed) into tools/_abl/; without it the script times whatever is there (run on the GPU box)."""
import ctypes as C, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
MASKS = [0, 1, 2, 4, 8, 16, 32, 64]
# --ab: whole-kernel variants timed ABBA in one process and compared with the first one: name -> (flags, source or None = product)
VARIANTS = {"product": ([], None), "variant": ([], os.path.join(HERE, "_abl", "attn_variant.hip"))}   # drop a modified attention.hip there
for a in sys.argv[1:]:
    if a.startswith("-D") or a.startswith("-f"):           # one variant per argument; "-DX=1,-fno-slp-vectorize" = several flags in one variant
        VARIANTS[a.lstrip("-").lower().replace("=", "").replace(",", "_")] = (a.split(","), None)
VARIANTS = {k: v for k, v in VARIANTS.items() if v[1] is None or os.path.exists(v[1])}
if "--build" in sys.argv and "--ab" in sys.argv:
    for name, (flags, src) in VARIANTS.items():
        out = os.path.join(HERE, "_abl", f"libattn_{name}.so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-I", os.path.join(ROOT, "zutis_amd/csrc")] + flags +
                              [src or os.path.join(ROOT, "zutis_amd/csrc/attention.hip"), os.path.join(ROOT, "zutis_amd/csrc/capi.hip"), "-o", out])
    sys.exit(0)
if "--build" in sys.argv:
    for m in MASKS:
        out = os.path.join(HERE, "_abl", f"libattn_{m}.so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", f"-DZH_ATTN_ABL={m}",
                               os.path.join(ROOT, "zutis_amd/csrc/attention.hip"), os.path.join(ROOT, "zutis_amd/csrc/capi.hip"), "-o", out])
    sys.exit(0)
import torch
dev = torch.device("cuda:0")
vp, l, i, f = C.c_void_p, C.c_long, C.c_int, C.c_float
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
if "--ab" in sys.argv:
    MASKS = list(VARIANTS)
names = {0: "full", 1: "-exp", 2: "-PV(+softmax DCE)", 4: "-S", 8: "-tiles-barriers", 16: "-barriers", 32: "barriers only", 64: "-LDS stores"}
for name, B, H, dh, Tq, Tk in [("enc", 32, 12, 64, 442, 442), ("cross", 32, 8, 96, 100, 1764), ("c4enc", 8, 12, 64, 1025, 1025), ("c5enc", 256, 16, 64, 577, 577),
                              ("selfmask", 1, 6, 64, 5505, 5505)]:
    D = H * dh
    X3 = "--x3" in sys.argv                    # split-pair kernels: [2, B, T, D] tensors, plane offset = B*T*D
    P = 2 if X3 else 1
    q = torch.randn(P, B, Tq, D, device=dev).half(); k = torch.randn(P, B, Tk, D, device=dev).half(); v = torch.randn(P, B, Tk, D, device=dev).half()
    if X3:
        q[1] *= 2 ** -11; k[1] *= 2 ** -11; v[1] *= 2 ** -11
    o = torch.empty(P, B, Tq, D, device=dev, dtype=torch.float16)
    pq, pk, po = (B * Tq * D, B * Tk * D, B * Tq * D) if X3 else (0, 0, 0)
    fns = {}
    for m in MASKS:
        L = C.CDLL(os.path.join(HERE, "_abl", f"libattn_{m}.so"))
        L.zh_attention_f16.restype = i
        L.zh_attention_f16.argtypes = [vp, l, l, vp, l, l, vp, l, l, vp, l, l, i, i, i, i, i, f, l, l, l, l, vp]
        s = torch.cuda.current_stream().cuda_stream
        fns[m] = (lambda L=L: L.zh_attention_f16(q.data_ptr(), D, Tq * D, k.data_ptr(), D, Tk * D, v.data_ptr(), D, Tk * D, o.data_ptr(), D, Tq * D,
                                                 B, H, Tq, Tk, dh, dh ** -0.5, pq, pk, pk, po, s))
        assert fns[m]() == 0
    if "--ab" in sys.argv:                                            # variants must agree with the first one
        outs = {}
        for m in MASKS:
            o.zero_(); fns[m](); torch.cuda.synchronize(); outs[m] = o.float().clone()
        diffs = " ".join(f"{m}:{float((outs[m] - outs[MASKS[0]]).abs().max()):.1e}" for m in MASKS[1:])
        print(f"{name:8s} max |diff| vs {MASKS[0]}: {diffs}  (|o| max {float(outs[MASKS[0]].abs().max()):.2f})")
    best = {m: 1e9 for m in MASKS}
    for order in (MASKS, MASKS[::-1], MASKS, MASKS[::-1]):           # ABBA: the first variant timed after a pause runs colder
        for m in order:
            best[m] = min(best[m], t(fns[m]))
    row = [f"{names.get(m, m)}:{best[m]:6.1f}" for m in MASKS]
    print(f"{name:8s}", "  ".join(row), "us")
