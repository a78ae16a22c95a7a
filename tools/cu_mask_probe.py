"""Feasibility probe: HIP streams restricted to a subset of the CUs (hipExtStreamCreateWithCUMask).  Times the encoder's QKV GEMM
on a 224-CU stream, a decoder-shaped GEMM chain on a 32-CU stream, and both at once — can the decoder's latency-bound kernels
hide behind the encoder's full-machine ones when each has its own CUs?"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zutis_amd import ops
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
torch.zeros(1, device=dev)
hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[sum(1 << j for j in range(32) if (32 * i + j) in bits) for i in range(8)])
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)
nD = int(sys.argv[1]) if len(sys.argv) > 1 else 32
SE = masked_stream(set(range(0, 256 - nD)))
SD = masked_stream(set(range(256 - nD, 256)))
M, K = 14144, 768
A = torch.randn(M, K, device=dev).half(); Wq = (torch.randn(2304, K, device=dev) * 0.03).half(); Oq = torch.empty(M, 2304, device=dev, dtype=torch.float16)
Wo = (torch.randn(768, K, device=dev) * 0.03).half(); X = torch.randn(M, 768, device=dev)
a = torch.randn(3200, 768, device=dev).half(); w = (torch.randn(768, 768, device=dev) * 0.03).half(); o = torch.empty(3200, 768, device=dev, dtype=torch.float16)
def enc(n):
    for _ in range(n):
        ops.gemm(A, Wq, Oq); ops.gemm(A, Wo, X, residual=X)
def dec(n):
    for _ in range(n): ops.gemm(a, w, o)
def timeit(fn_e, fn_d, se, sd):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    d0, d1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if fn_e:
        with torch.cuda.stream(se): e0.record(); fn_e(); e1.record()
    if fn_d:
        with torch.cuda.stream(sd): d0.record(); fn_d(); d1.record()
    torch.cuda.synchronize()
    return (e0.elapsed_time(e1) if fn_e else 0.0, d0.elapsed_time(d1) if fn_d else 0.0)
S0, S1 = torch.cuda.Stream(), torch.cuda.Stream()
for _ in range(2):
    timeit(lambda: enc(5), lambda: dec(20), SE, SD); timeit(lambda: enc(5), lambda: dec(20), S0, S1)
NE, ND = 40, 240
print(f"D stream has {nD} CUs")
print("encoder pair x%d alone : unmasked %.2f ms, masked(%d CUs) %.2f ms" % (NE, timeit(lambda: enc(NE), None, S0, S1)[0], 256 - nD, timeit(lambda: enc(NE), None, SE, SD)[0]))
print("decoder gemm x%d alone: unmasked %.2f ms, masked(%d CUs) %.2f ms" % (ND, timeit(None, lambda: dec(ND), S0, S1)[1], nD, timeit(None, lambda: dec(ND), SE, SD)[1]))
te, td = timeit(lambda: enc(NE), lambda: dec(ND), S0, S1); print("both, two plain streams : enc %.2f ms, dec %.2f ms" % (te, td))
te, td = timeit(lambda: enc(NE), lambda: dec(ND), SE, SD); print("both, CU-masked streams : enc %.2f ms, dec %.2f ms" % (te, td))
