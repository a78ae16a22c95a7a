// Developer micro-benchmark: what a CU's LDS-DMA stream sustains when it is PIPELINED like the GEMM's K loop (a ring of slots, counted
// vmcnt waits, no LDS reads, no MFMAs) as a function of the piece shape (16 rows x 64 B = BK 32, 8 x 128 B = BK 64, 4 x 256 B), the number
// of waves that issue and the slices in flight.  Each workgroup streams ROWS rows of a row-major matrix (row stride 1536 B) along K,
// as a GEMM tile does; workgroups share rows the way a tile grid does (A panel by tile row, W panel by tile column).
//   hipcc --offload-arch=gfx950 -O3 -o dma_stream dma_stream.hip && ./dma_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// PB: bytes per row piece; NW: waves; ROWS: rows per slice; DEPTH: slices in flight; SLOTS = DEPTH + 1
template <int PB, int NW, int ROWS, int DEPTH>
__global__ __launch_bounds__(64 * NW) void k(const char* a, const char* w, int a_rows, int w_rows, int row_bytes, int nbn, int* sink) {
  constexpr int RPI = 1024 / PB;                 // rows per wave instruction (1 KiB pieces)
  constexpr int PIECES = ROWS / RPI;             // pieces per slice
  constexpr int NP = (PIECES + NW - 1) / NW;     // per wave
  constexpr int SLOTS = DEPTH + 1;
  __shared__ __attribute__((aligned(16))) char lds[SLOTS * ROWS * PB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tm = blockIdx.x / nbn, tn = blockIdx.x % nbn;
  const int r = lane / (PB / 16), c = lane % (PB / 16);
  const char* src[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    int pc = wave + i * NW; pc = pc < PIECES ? pc : PIECES - 1;
    const int row = pc * RPI + r;                // 0 .. ROWS-1: first half from A (tile row tm), second half from W (tile column tn)
    const int half = ROWS / 2;
    src[i] = row < half ? a + (long)((tm * half + row) % a_rows) * row_bytes + c * 16
                        : w + (long)((tn * half + row - half) % w_rows) * row_bytes + c * 16;
  }
  const int nslice = row_bytes / PB;
  auto issue = [&](int s) {
    char* dst = lds + (s % SLOTS) * (ROWS * PB);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      int pc = wave + i * NW; pc = pc < PIECES ? pc : PIECES - 1;
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(src[i] + (long)s * PB), (lds_ptr_t)(dst + pc * 1024), 16, 0, 0);
    }
  };
  for (int s = 0; s < DEPTH && s < nslice; ++s) issue(s);
  for (int s = 0; s < nslice; ++s) {
    wait_vm<(DEPTH - 1) * NP>();                 // slice s landed (for this wave's pieces)
    if (s + DEPTH < nslice) issue(s + DEPTH);
  }
  wait_vm<0>();
  if (lds[threadIdx.x] == 123 && sink) sink[0] = 1;
}

template <int PB, int NW, int ROWS, int DEPTH>
void run(const char* a, const char* w, int a_rows, int w_rows, int rb, int nbm, int nbn, int* sink, const char* what) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = nbm * nbn;
  hipLaunchKernelGGL((k<PB, NW, ROWS, DEPTH>), dim3(grid), dim3(64 * NW), 0, 0, a, w, a_rows, w_rows, rb, nbn, sink);
  hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<PB, NW, ROWS, DEPTH>), dim3(grid), dim3(64 * NW), 0, 0, a, w, a_rows, w_rows, rb, nbn, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps, bytes = (double)ROWS * rb;
  printf("%-28s piece %3d B, %2d waves, %3d rows/slice, %d in flight, grid %3d: %6.1f us per launch, %5.1f GB/s per workgroup, %5.2f TB/s chip\n",
         what, PB, NW, ROWS, DEPTH, grid, us, bytes / us / 1e3, bytes * grid / us / 1e6);
}

int main() {
  const int rb = 1536 * 2;                       // K = 768 halves x (hi, lo) planes side by side: 3072 B of row per "row pair"
  const int a_rows = 1280, w_rows = 2304;        // QKV at batch 1: A 1201 rows (10 tiles of 128), W 2304 rows (24 tiles of 96 / 18 of 128)
  char *a, *w; int* sink;
  hipMalloc(&a, (size_t)a_rows * rb); hipMemset(a, 1, (size_t)a_rows * rb);
  hipMalloc(&w, (size_t)w_rows * rb); hipMemset(w, 1, (size_t)w_rows * rb);
  hipMalloc(&sink, 4);
  // rows per slice here are ROW PAIRS x 1 (the planes sit side by side in a 3072-B row): 128 + 96 = 224 "rows" of 2 x 64 B ... keep it simple:
  // ROWS counts 64-B (or PB-byte) row pieces per slice as the GEMM stages them: 2 x (128 + 96) = 448 for the 128 x 96 x3 tile.
  run<64, 8, 448, 2>(a, w, a_rows, w_rows, rb, 10, 24, sink, "128x96 x3, 3 slots");
  run<64, 8, 448, 4>(a, w, a_rows, w_rows, rb, 10, 24, sink, "128x96 x3, 5 slots");
  run<64, 4, 448, 4>(a, w, a_rows, w_rows, rb, 10, 24, sink, "same, 4 waves");
  run<64, 16, 448, 4>(a, w, a_rows, w_rows, rb, 10, 24, sink, "same, 16 waves");
  run<128, 8, 448, 1>(a, w, a_rows, w_rows, rb, 10, 24, sink, "BK 64 (128-B pieces), 2 slots");
  run<128, 8, 224, 4>(a, w, a_rows, w_rows, rb, 19, 24, sink, "BK 64, 64x48-ish tile, 5 slots");
  run<128, 16, 224, 4>(a, w, a_rows, w_rows, rb, 19, 24, sink, "same, 16 waves");
  run<256, 8, 224, 1>(a, w, a_rows, w_rows, rb, 19, 24, sink, "BK 128 (256-B pieces), 2 slots");
  run<64, 8, 512, 4>(a, w, a_rows, w_rows, rb, 10, 18, sink, "128x128 x3, 5 slots");
  run<64, 4, 256, 6>(a, w, a_rows, w_rows, rb, 19, 36, sink, "64x64 x3, 7 slots, 684 wgs");
  run<64, 4, 256, 6>(a, w, a_rows, w_rows, rb, 16, 16, sink, "64x64 x3, 7 slots, 256 wgs");
  run<64, 8, 448, 4>(a, w, a_rows, w_rows, rb, 8, 8, sink, "128x96, 64 wgs only");
  run<64, 8, 448, 4>(a, w, a_rows, w_rows, rb, 1, 1, sink, "128x96, ONE wg");
  return 0;
}
