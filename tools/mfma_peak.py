"""Developer tool: sustained MFMA rate of the two fp16 MFMA shapes at 1/2/4 waves per SIMD (tools/ab/mfma_peak.hip)."""
import sys, os, time, ctypes as C, torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab", "libmfma_peak.so"))
lib.run_mfma_peak.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
out = torch.zeros(4096 * 256, device="cuda")
for shape, flop in ((16, 16 * 16 * 32 * 2 * 16), (32, 32 * 32 * 16 * 2 * 8)):
    for blocks in (256, 512, 1024):          # 1, 2, 4 waves per SIMD
        iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
        for _ in range(2): lib.run_mfma_peak(out.data_ptr(), shape, blocks, iters, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize(); t = time.perf_counter()
        lib.run_mfma_peak(out.data_ptr(), shape, blocks, iters, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        print(f"mfma {shape}: {blocks//256} wave(s)/SIMD: {blocks * 4 * iters * flop / dt / 1e12:.0f} TFLOP/s ({dt*1e3:.1f} ms)")
