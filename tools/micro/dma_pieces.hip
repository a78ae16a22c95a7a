// Developer micro-benchmark: LDS-DMA (global_load_lds_dwordx4) rate per CU as a function of the PIECE a wave instruction gathers:
// 16 rows x 64 B (the GEMM's BK = 32 slices), 8 rows x 128 B (BK = 64), 4 x 256 B, 1 x 1 KiB, from an L2-resident row-major matrix
// (row stride 1536 B = K 768 halves).  hipcc --offload-arch=gfx950 -O3 -o dma_pieces dma_pieces.hip && ./dma_pieces
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
template <int PB>   // bytes per row piece
__global__ __launch_bounds__(256) void k(const char* src, int rows, int row_bytes, int iters, int* sink) {
  __shared__ __attribute__((aligned(16))) char lds[4 * 8 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int CPR = PB / 16, RPI = 64 / CPR;                 // chunks per row piece, rows per instruction
  const int r = lane / CPR, c = lane % CPR;
  const int nb = rows / (RPI * 4);                             // row blocks (4 waves x RPI rows)
  int acc = 0;
  for (int it = 0; it < iters; ++it) {
    const int rb = (blockIdx.x + it) % nb;
    const char* base = src + (long)(rb * RPI * 4 + wave * RPI + r) * row_bytes + c * 16;
    for (int kk = 0; kk + PB * 8 <= row_bytes; kk += PB * 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(base + kk + j * PB), (lds_ptr_t)(lds + (wave * 8 + j) * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      acc += lds[(wave * 8) * 1024 + lane * 4];
    }
  }
  if (acc == 12345) sink[0] = acc;
}
template <int PB> void run(const char* src, int rows, int rb, int* sink, int cus) {
  const int iters = 64;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu) {
    const int grid = cus * blocks_per_cu;
    k<PB><<<grid, 256>>>(src, rows, rb, 4, sink);
    hipEventRecord(e0);
    k<PB><<<grid, 256>>>(src, rows, rb, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * iters * (rb / (PB * 8)) * 4 * 8 * 1024;
    printf("piece %4d B x %2d rows, %d block(s)/CU: %7.1f GB/s per CU, %6.2f TB/s chip\n", PB, 64 / (PB / 16), blocks_per_cu, bytes / ms / 1e6 / cus, bytes / ms / 1e9);
  }
}
int main() {
  const int rows = 3200, rb = 1536 * 2;        // 3200 rows x 3072 B (two planes side by side is close enough): 9.8 MB, L2 / MALL resident
  char* src; int* sink;
  hipMalloc(&src, (size_t)rows * rb); hipMemset(src, 1, (size_t)rows * rb); hipMalloc(&sink, 4);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  run<64>(src, rows, rb, sink, cus); run<128>(src, rows, rb, sink, cus); run<256>(src, rows, rb, sink, cus); run<1024>(src, rows, rb, sink, cus);
  return 0;
}
