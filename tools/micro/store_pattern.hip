// Developer microbenchmark: HBM write rate of a [M x N] fp16 matrix written tile by tile (256 x 256 tiles, one 512-thread block per
// tile, 256 blocks resident) with different per-instruction shapes:  mode 0: 8 B per lane, 4 rows x 128 B per wave instruction
// (the split-pair GEMM epilogue: each wave owns a 64-column slab);  mode 1: 8 B per lane, 1 row x 512 B;  mode 2: 16 B per lane,
// 2 rows x 512 B;  mode 3: 16 B per lane, 8 rows x 128 B (the fp16 GEMM epilogue).   hipcc --offload-arch=gfx950 -O3 store_pattern.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half_t;
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ __launch_bounds__(512) void k(half_t* out, long ld, int nbn, int planes, long plane) {
  const int tile = blockIdx.x, tm = tile / nbn, tn = tile % nbn;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wr = wave >> 2, wc = wave & 3;
  half_t* base = out + (long)tm * 256 * ld + tn * 256;
  const half4_t v4 = {(half_t)1, (half_t)2, (half_t)3, (half_t)4};
  const half8_t v8 = {(half_t)1, (half_t)2, (half_t)3, (half_t)4, (half_t)5, (half_t)6, (half_t)7, (half_t)8};
  for (int pl = 0; pl < planes; ++pl) {
    half_t* b = base + pl * plane;
    if (MODE == 0) {          // wave owns rows [wr*128, +128) x cols [wc*64, +64): instruction = 4 rows x 16 lanes x 8 B
      for (int it = 0; it < 32; ++it) { const int row = wr * 128 + it * 4 + (lane >> 4), col = wc * 64 + (lane & 15) * 4; *(half4_t*)(b + (long)row * ld + col) = v4; }
    } else if (MODE == 1) {   // wave owns rows [wave*32, +32) x all 256 cols: instruction = 1 row x 64 lanes x 8 B
      for (int it = 0; it < 32; ++it) { const int row = wave * 32 + it, col = lane * 4; *(half4_t*)(b + (long)row * ld + col) = v4; }
    } else if (MODE == 2) {   // 16 B per lane: 2 rows x 32 lanes x 16 B
      for (int it = 0; it < 16; ++it) { const int row = wave * 32 + it * 2 + (lane >> 5), col = (lane & 31) * 8; *(half8_t*)(b + (long)row * ld + col) = v8; }
    } else {                  // 16 B per lane, wave owns a 64-column slab: 8 rows x 8 lanes x 16 B
      for (int it = 0; it < 16; ++it) { const int row = wr * 128 + it * 8 + (lane >> 3), col = wc * 64 + (lane & 7) * 8; *(half8_t*)(b + (long)row * ld + col) = v8; }
    }
  }
}
template <int MODE> static void run(half_t* out, long M, long N, int planes) {
  const int nbm = M / 256, nbn = N / 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(nbm * nbn), dim3(512), 0, 0, out, N, nbn, planes, M * N);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k<MODE>, dim3(nbm * nbn), dim3(512), 0, 0, out, N, nbn, planes, M * N);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("mode %d planes %d: %.1f us per pass, %.2f TB/s\n", MODE, planes, ms * 100.0, (double)M * N * 2 * planes / (ms / 10 * 1e-3) / 1e12);
}
int main() {
  const long M = 56320, N = 4608;
  half_t* out; hipMalloc(&out, M * N * 2 * 2);
  for (int planes = 1; planes <= 2; ++planes) { run<0>(out, M, N, planes); run<1>(out, M, N, planes); run<2>(out, M, N, planes); run<3>(out, M, N, planes); }
  return 0;
}
