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
// The lo halves come from v_fma_mix{lo,hi}_f16 — ONE instruction per value: it reads hi as fp16 out of the packed pair, forms
// y - hi in fp32 (exact: the difference has at most 13 significant bits) and rounds to fp16, i.e. the same number as
// (half)(y - (float)hi) for 3 instructions (v_cvt_f32_f16, v_sub_f32, half a v_cvt_pk_f16_f32).
__device__ __forceinline__ unsigned zh_lo_pair(unsigned hpair, float y0, float y1) {
  unsigned l;
  asm("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %0, -%1, 1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(l) : "v"(hpair), "v"(y0), "v"(y1));
  return l;
}
__device__ __forceinline__ void zh_store_h4(half_t* p, long lo_plane, f32x4 y) {
  typedef unsigned zh_u32x2 __attribute__((ext_vector_type(2)));
  const half2_t h01 = {(half_t)y[0], (half_t)y[1]}, h23 = {(half_t)y[2], (half_t)y[3]};
  zh_u32x2 h = {__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
  asm volatile("" : "+v"(h));
  *(zh_u32x2*)p = h;
  if (lo_plane) {
    const zh_u32x2 l = {zh_lo_pair(h[0], y[0], y[1]), zh_lo_pair(h[1], y[2], y[3])};
    *(zh_u32x2*)(p + lo_plane) = l;
  }
}
// 8 consecutive values as a split pair: two 16-byte stores (hi plane, lo plane)
__device__ __forceinline__ void zh_store_h8(half_t* p, long lo_plane, f32x4 a, f32x4 b) {
  typedef unsigned zh_u32x4 __attribute__((ext_vector_type(4)));
  const half2_t h0 = {(half_t)a[0], (half_t)a[1]}, h1 = {(half_t)a[2], (half_t)a[3]}, h2 = {(half_t)b[0], (half_t)b[1]}, h3 = {(half_t)b[2], (half_t)b[3]};
  zh_u32x4 h = {__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1), __builtin_bit_cast(unsigned, h2), __builtin_bit_cast(unsigned, h3)};
  asm volatile("" : "+v"(h));
  *(zh_u32x4*)p = h;
  const zh_u32x4 l = {zh_lo_pair(h[0], a[0], a[1]), zh_lo_pair(h[1], a[2], a[3]), zh_lo_pair(h[2], b[0], b[1]), zh_lo_pair(h[3], b[2], b[3])};
  *(zh_u32x4*)(p + lo_plane) = l;
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
