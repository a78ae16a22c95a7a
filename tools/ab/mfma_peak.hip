// Developer microbenchmark: sustained MFMA rate (power-limited clock included) for the two fp16 shapes, 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_peak(float* out, int iters, float seed) {
  half8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + threadIdx.x * 0.001f + i); b[i] = (_Float16)(seed * 0.5f - i); }
  if (SHAPE == 16) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  }
}

extern "C" int run_mfma_peak(float* out, int shape, int blocks, int iters, void* stream) {
  if (shape == 16) hipLaunchKernelGGL(mfma_peak<16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters, 1.0f);
  else hipLaunchKernelGGL(mfma_peak<32>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters, 1.0f);
  return (int)hipGetLastError();
}
