// Shared device/host helpers for libzutis_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define ZH_OK 0
#define ZH_ERR_ARG (-1)
#define ZH_ERR_HIP (-2)
#define ZH_ERR_WORKSPACE (-3)

// thread-local last-error text, returned by zh_last_error()
void zh_set_error(const char* fmt, ...);

#define ZH_CHECK_ARG(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      zh_set_error(__VA_ARGS__);           \
      return ZH_ERR_ARG;                   \
    }                                      \
  } while (0)

#define ZH_CHECK_LAUNCH(name)                                              \
  do {                                                                     \
    hipError_t e__ = hipGetLastError();                                    \
    if (e__ != hipSuccess) {                                               \
      zh_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return ZH_ERR_HIP;                                                   \
    }                                                                      \
  } while (0)

static inline int zh_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- wave64 reductions ------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// fp16 store of 4 consecutive values, optionally as a split pair for the f16x3 GEMM mode (gemm_x3.hip): hi = f16(y) at p,
// lo = f16(y - hi) at p + lo_plane.  lo_plane == 0: hi only (the plain fp16 tensor every other consumer reads).
// NOTE the asm barrier: without it hipcc converts y -> fp16 TWICE (v_cvt_pk_f16_f32 for the stored vector, v_cvt_f16_f32 for
// the value subtracted below) and on gfx950 the two instructions do not agree on every input (seen on exact rounding ties,
// ~2^-13 of all values): hi + lo was then off by one fp16 ulp.  The barrier makes `h` opaque, so lo is computed from the very
// bits that are stored.
__device__ __forceinline__ void zh_store_h4(half_t* p, long lo_plane, f32x4 y) {
  half4_t h = {(half_t)y[0], (half_t)y[1], (half_t)y[2], (half_t)y[3]};
  asm volatile("" : "+v"(h));
  *(half4_t*)p = h;
  if (lo_plane) {
    const half4_t l = {(half_t)(y[0] - (float)h[0]), (half_t)(y[1] - (float)h[1]), (half_t)(y[2] - (float)h[2]),
                       (half_t)(y[3] - (float)h[3])};
    *(half4_t*)(p + lo_plane) = l;
  }
}
__device__ __forceinline__ void zh_store_h1(half_t* p, long lo_plane, float y) {
  half_t h = (half_t)y;
  asm volatile("" : "+v"(h));
  *p = h;
  if (lo_plane) p[lo_plane] = (half_t)(y - (float)h);
}

// activation codes shared by the GEMM epilogue (include/zutis_hip.h ZH_ACT_*)
#define ZH_ACT_NONE 0
#define ZH_ACT_QUICKGELU 1   // x * sigmoid(1.702 x)      networks/clip_arch.py:295-297
#define ZH_ACT_RELU 2        // networks/zutis.py:546-549, networks/transformer.py:289
#define ZH_ACT_SIGMOID 3     // networks/zutis.py:209
#define ZH_ACT_GELU_ERF 4    // nn.GELU, networks/selfmask/vision_transformer.py:79

__device__ __forceinline__ float zh_act(float x, int act) {
  // v_exp_f32 / v_rcp_f32 (1 ulp) instead of IEEE expf + division: the epilogue of the 14144x3072 c_fc GEMM spends
  // 43 M activations per call; results differ from the libm form by < 2e-7 relative (tests: 2e-4 abs on O(1) values).
  switch (act) {
    case ZH_ACT_QUICKGELU: return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x));
    case ZH_ACT_RELU: return fmaxf(x, 0.0f);
    case ZH_ACT_SIGMOID: return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
    case ZH_ACT_GELU_ERF: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
    default: return x;
  }
}
