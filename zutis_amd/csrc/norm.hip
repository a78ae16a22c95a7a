// Row reductions: LayerNorm family, per-image global LN + per-pixel L2, row L2 normalise (gfx950).
// HBM-bound; one 64-lane wave per row, 16-byte loads, row kept in registers between the two moments
// (no re-read), fp32 statistics (two-pass mean / centred variance, biased — nn.LayerNorm semantics).
#include "common.h"

#define LN_MAXV 4  // float4 per lane -> D <= 1024

struct LnRow {
  f32x4 v[LN_MAXV];
};

// The engine's status word (optional `status` argument of the LayerNorm family): bit 1 is OR-ed in when a row's variance is not a finite
// number — an inf or a NaN reached the residual stream.  That is how an fp16 split pair leaving its range shows: |activation| >= 65504
// stores hi = inf, lo = -inf / NaN, the next product is a NaN, and every path of the model (residual adds, attention over an image's
// tokens) leads into a LayerNorm.  One compare per row here, nothing in the GEMM / attention epilogues; the host reads the word at its
// next synchronisation and raises (no fallback).
#define ZH_STATUS_NONFINITE 2

// y = (x - mean) * rstd, in place in registers. nv = D/4 float4 per row.  Returns false when the variance is inf / NaN.
__device__ __forceinline__ bool ln_normalize(LnRow& r, int nv, int lane, int D, float eps) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j)
    if (lane + 64 * j < nv) s += (r.v[j][0] + r.v[j][1]) + (r.v[j][2] + r.v[j][3]);
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j)
    if (lane + 64 * j < nv) {
      r.v[j] -= mean;
      q += (r.v[j][0] * r.v[j][0] + r.v[j][1] * r.v[j][1]) + (r.v[j][2] * r.v[j][2] + r.v[j][3] * r.v[j][3]);
    }
  const float var = wave_sum(q) / (float)D;
  const float rstd = rsqrtf(var + eps);
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) r.v[j] *= rstd;
  return var < INFINITY;                                    // false for inf and for NaN
}
__device__ __forceinline__ void zh_raise_nonfinite(int* status, bool ok, int lane) {
  if (status && !ok && lane == 0) atomicOr(status, ZH_STATUS_NONFINITE);
}

struct LnArgs {
  const float* x; long in_group_rows, in_group_stride, in_offset;  // in_row = (r / group_rows)*stride + offset + r % group_rows
  long out_group_rows, out_group_stride, out_offset;                // same mapping for every output
  const float* gamma; const float* beta;
  float* out_f32; half_t* out_f16;
  half_t* out_f16_plus; const float* add; int add_rows;             // out_f16_plus = fp16(y + add[r % add_rows])
  float* out_f32_plus;                                              // optional fp32 copy of y + add
  int rows, D; float eps;
  long lo_plane;                                                    // != 0: fp16 outputs are split pairs (lo half at + lo_plane)
  int* status;                                                      // optional status word (ZH_STATUS_NONFINITE)
};

__global__ __launch_bounds__(256) void layernorm_kernel(LnArgs p) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= p.rows) return;
  const int nv = p.D >> 2;
  const long in_row = (r / p.in_group_rows) * p.in_group_stride + p.in_offset + (r % p.in_group_rows);
  const f32x4* xp = (const f32x4*)(p.x + in_row * p.D);
  LnRow row;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j)
    if (lane + 64 * j < nv) row.v[j] = xp[lane + 64 * j];
  zh_raise_nonfinite(p.status, ln_normalize(row, nv, lane, p.D, p.eps), lane);
  const long ob = ((r / p.out_group_rows) * p.out_group_stride + p.out_offset + (r % p.out_group_rows)) * p.D;
  const long ab = p.add ? (long)(r % p.add_rows) * p.D : 0;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    const int c = lane + 64 * j;
    if (c < nv) {
      f32x4 y = row.v[j];
      if (p.gamma) y = y * ((const f32x4*)p.gamma)[c] + ((const f32x4*)p.beta)[c];
      if (p.out_f32) ((f32x4*)(p.out_f32 + ob))[c] = y;
      if (p.out_f16) zh_store_h4(p.out_f16 + ob + 4 * c, p.lo_plane, y);
      if (p.add) {
        f32x4 z = y + ((const f32x4*)(p.add + ab))[c];
        if (p.out_f16_plus) zh_store_h4(p.out_f16_plus + ob + 4 * c, p.lo_plane, z);
        if (p.out_f32_plus) ((f32x4*)(p.out_f32_plus + ob))[c] = z;
      }
    }
  }
}

extern "C" int zh_layernorm_f32(const float* x, long in_group_rows, long in_group_stride, long in_offset,
                                long out_group_rows, long out_group_stride, long out_offset,
                                const float* gamma, const float* beta, float eps,
                                float* out_f32, void* out_f16, void* out_f16_plus, float* out_f32_plus,
                                const float* add, int add_rows, int rows, int D, long lo_plane, int* status, hipStream_t stream) {
  ZH_CHECK_ARG(x && rows > 0, "zh_layernorm_f32: bad input");
  ZH_CHECK_ARG(lo_plane % 4 == 0, "zh_layernorm_f32: lo_plane must be a multiple of 4 halves");
  ZH_CHECK_ARG(D % 4 == 0 && D <= 256 * LN_MAXV && D > 0, "zh_layernorm_f32: D=%d must be a multiple of 4 and <= %d", D, 256 * LN_MAXV);
  ZH_CHECK_ARG((gamma == nullptr) == (beta == nullptr), "zh_layernorm_f32: gamma and beta must both be given or both null");
  ZH_CHECK_ARG(in_group_rows > 0 && out_group_rows > 0, "zh_layernorm_f32: group_rows must be > 0");
  ZH_CHECK_ARG(!(out_f16_plus || out_f32_plus) || (add && add_rows > 0), "zh_layernorm_f32: *_plus outputs need add/add_rows");
  LnArgs p{x, in_group_rows, in_group_stride, in_offset, out_group_rows, out_group_stride, out_offset, gamma, beta, out_f32, (half_t*)out_f16,
           (half_t*)out_f16_plus, add, add_rows, out_f32_plus, rows, D, eps, lo_plane, status};
  hipLaunchKernelGGL(layernorm_kernel, dim3(zh_cdiv(rows, 4)), dim3(256), 0, stream, p);
  ZH_CHECK_LAUNCH("zh_layernorm_f32");
  return ZH_OK;
}

// ---- split-K combine + bias + residual + LayerNorm (+ a second, chained LayerNorm) in one pass over the row.
//   x = ((parts[0] + parts[1] + ... ) + bias) + residual           (the order of the GEMM epilogue it replaces)
//   out_sum = x;  y = LN(x; gamma, beta, eps) -> out_f32 / out_f16 at out_row(r);  z = LN(y; gamma2, beta2, eps2) -> out2_* at out2_row(r)
// Batch-1 evaluation (configs/*: val batch_size 1; trainer.py:328-345): the N = 768 GEMMs of a block (out_proj, c_proj,
// clip_arch.py:318-321) have 60 tiles for 256 CUs unless K is split; their fp32 partial planes are summed HERE, by the kernel
// that had to read the sum anyway (the next LayerNorm), in plane order — deterministic.  The decoder's norm3 -> decoder.norm
// pair (transformer.py:140-150,291) is the chained form.
struct SumLnArgs {
  const float* parts; int n_parts; long part_stride;
  const float* bias; const float* residual; float* out_sum;
  const float* gamma; const float* beta; float eps;
  float* out_f32; half_t* out_f16; long lo_plane;
  long og_rows, og_stride, og_offset; int skip_first;              // out_row(r) = (r / og_rows) * og_stride + og_offset + r % og_rows; skip r % og_rows == 0
  const float* gamma2; const float* beta2; float eps2;
  float* out2_f32; half_t* out2_f16; long lo_plane2;
  long og2_rows, og2_stride, og2_offset;
  int rows, D;
  int* status;
};

// NPARTS > 0: the plane count as a template parameter — every plane's loads (and the bias / residual ones) are requested before the first add
// (a run-time plane loop issues one plane, waits, adds: four dependent latencies for c_proj's four planes, 10.4 us for 1201 rows where the
// plain LayerNorm takes 5.4); the additions keep plane order.  NPARTS = 0: any count, the loop.
template <int NPARTS>
__global__ __launch_bounds__(256) void sum_layernorm_kernel(SumLnArgs p) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= p.rows) return;
  const int nv = p.D >> 2;
  LnRow row;
  if constexpr (NPARTS > 0) {
    f32x4 pv[NPARTS][LN_MAXV], bv[LN_MAXV], rv[LN_MAXV];
#pragma unroll
    for (int s = 0; s < NPARTS; ++s) {
      const f32x4* ps = (const f32x4*)(p.parts + (long)s * p.part_stride + r * p.D);
#pragma unroll
      for (int j = 0; j < LN_MAXV; ++j)
        if (lane + 64 * j < nv) pv[s][j] = ps[lane + 64 * j];
    }
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
      const int c = lane + 64 * j;
      if (c < nv) {
        bv[j] = p.bias ? ((const f32x4*)p.bias)[c] : (f32x4){0.f, 0.f, 0.f, 0.f};
        rv[j] = p.residual ? ((const f32x4*)(p.residual + r * p.D))[c] : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
      const int c = lane + 64 * j;
      if (c < nv) {
        f32x4 x = pv[0][j];
#pragma unroll
        for (int s = 1; s < NPARTS; ++s) x += pv[s][j];
        if (p.bias) x += bv[j];
        if (p.residual) x += rv[j];
        row.v[j] = x;
        if (p.out_sum) ((f32x4*)(p.out_sum + r * p.D))[c] = x;
      }
    }
  } else {
    const f32x4* p0 = (const f32x4*)(p.parts + r * p.D);
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j)
      if (lane + 64 * j < nv) row.v[j] = p0[lane + 64 * j];
    for (int s = 1; s < p.n_parts; ++s) {
      const f32x4* ps = (const f32x4*)(p.parts + (long)s * p.part_stride + r * p.D);
#pragma unroll
      for (int j = 0; j < LN_MAXV; ++j)
        if (lane + 64 * j < nv) row.v[j] += ps[lane + 64 * j];
    }
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
      const int c = lane + 64 * j;
      if (c < nv) {
        if (p.bias) row.v[j] += ((const f32x4*)p.bias)[c];
        if (p.residual) row.v[j] += ((const f32x4*)(p.residual + r * p.D))[c];
        if (p.out_sum) ((f32x4*)(p.out_sum + r * p.D))[c] = row.v[j];
      }
    }
  }
  if (!p.gamma) return;
  zh_raise_nonfinite(p.status, ln_normalize(row, nv, lane, p.D, p.eps), lane);
  const long g = r / p.og_rows, w = r % p.og_rows;
  const bool emit = !(p.skip_first && w == 0);
  const long ob = (g * p.og_stride + p.og_offset + w) * p.D;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    const int c = lane + 64 * j;
    if (c < nv) {
      row.v[j] = row.v[j] * ((const f32x4*)p.gamma)[c] + ((const f32x4*)p.beta)[c];
      if (emit) {
        if (p.out_f32) ((f32x4*)(p.out_f32 + ob))[c] = row.v[j];
        if (p.out_f16) zh_store_h4(p.out_f16 + ob + 4 * c, p.lo_plane, row.v[j]);
      }
    }
  }
  if (!p.gamma2) return;
  ln_normalize(row, nv, lane, p.D, p.eps2);
  const long ob2 = ((r / p.og2_rows) * p.og2_stride + p.og2_offset + (r % p.og2_rows)) * p.D;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    const int c = lane + 64 * j;
    if (c < nv) {
      const f32x4 z = row.v[j] * ((const f32x4*)p.gamma2)[c] + ((const f32x4*)p.beta2)[c];
      if (p.out2_f32) ((f32x4*)(p.out2_f32 + ob2))[c] = z;
      if (p.out2_f16) zh_store_h4(p.out2_f16 + ob2 + 4 * c, p.lo_plane2, z);
    }
  }
}

extern "C" int zh_sum_layernorm_f32(const float* parts, int n_parts, long part_stride, const float* bias, const float* residual,
                                    float* out_sum, const float* gamma, const float* beta, float eps,
                                    float* out_f32, void* out_f16, long lo_plane,
                                    long out_group_rows, long out_group_stride, long out_offset, int skip_first_in_group,
                                    const float* gamma2, const float* beta2, float eps2, float* out2_f32, void* out2_f16, long lo_plane2,
                                    long out2_group_rows, long out2_group_stride, long out2_offset,
                                    int rows, int D, int* status, hipStream_t stream) {
  ZH_CHECK_ARG(parts && n_parts >= 1 && rows > 0, "zh_sum_layernorm_f32: bad input");
  ZH_CHECK_ARG(n_parts == 1 || part_stride >= (long)rows * D, "zh_sum_layernorm_f32: part_stride %ld < rows * D", part_stride);
  ZH_CHECK_ARG(D % 4 == 0 && D <= 256 * LN_MAXV && D > 0, "zh_sum_layernorm_f32: D=%d must be a multiple of 4 and <= %d", D, 256 * LN_MAXV);
  ZH_CHECK_ARG(part_stride % 4 == 0 && lo_plane % 4 == 0 && lo_plane2 % 4 == 0, "zh_sum_layernorm_f32: strides / planes must be multiples of 4 elements");
  ZH_CHECK_ARG((gamma == nullptr) == (beta == nullptr) && (gamma2 == nullptr) == (beta2 == nullptr), "zh_sum_layernorm_f32: gamma and beta go together");
  ZH_CHECK_ARG(gamma || !(out_f32 || out_f16 || gamma2), "zh_sum_layernorm_f32: LayerNorm outputs need gamma / beta");
  ZH_CHECK_ARG(gamma || out_sum, "zh_sum_layernorm_f32: nothing to write");
  ZH_CHECK_ARG(!gamma || (out_group_rows > 0 && (out_f32 || out_f16 || gamma2)), "zh_sum_layernorm_f32: first LayerNorm needs an output and out_group_rows > 0");
  ZH_CHECK_ARG(!gamma2 || (out2_group_rows > 0 && (out2_f32 || out2_f16)), "zh_sum_layernorm_f32: second LayerNorm needs an output and out2_group_rows > 0");
  SumLnArgs p{parts, n_parts, part_stride, bias, residual, out_sum, gamma, beta, eps, out_f32, (half_t*)out_f16, lo_plane,
              out_group_rows, out_group_stride, out_offset, skip_first_in_group, gamma2, beta2, eps2, out2_f32, (half_t*)out2_f16, lo_plane2,
              out2_group_rows, out2_group_stride, out2_offset, rows, D, status};
  const dim3 grid(zh_cdiv(rows, 4));
  if (n_parts == 1) hipLaunchKernelGGL(sum_layernorm_kernel<1>, grid, dim3(256), 0, stream, p);
  else if (n_parts == 2) hipLaunchKernelGGL(sum_layernorm_kernel<2>, grid, dim3(256), 0, stream, p);
  else if (n_parts == 4) hipLaunchKernelGGL(sum_layernorm_kernel<4>, grid, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(sum_layernorm_kernel<0>, grid, dim3(256), 0, stream, p);
  ZH_CHECK_LAUNCH("zh_sum_layernorm_f32");
  return ZH_OK;
}

// ---- token assembly + ln_pre: networks/clip_arch.py:384-397
//   t[b,0] = class_embedding + pos[0];  t[b,1+i] = patch[b,i] + pos[1+i];  X = LN(t)
struct AsmArgs {
  const float* patch; const float* cls; const float* pos; const float* gamma; const float* beta;
  float* out; int B, T, D; float eps;
};

__global__ __launch_bounds__(256) void assemble_ln_kernel(AsmArgs p) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= (long)p.B * p.T) return;
  const int nv = p.D >> 2;
  const int b = (int)(r / p.T), t = (int)(r % p.T);
  const f32x4* src = t == 0 ? (const f32x4*)p.cls : (const f32x4*)(p.patch + ((long)b * (p.T - 1) + (t - 1)) * p.D);
  const f32x4* pos = (const f32x4*)(p.pos + (long)t * p.D);
  LnRow row;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j)
    if (lane + 64 * j < nv) row.v[j] = src[lane + 64 * j] + pos[lane + 64 * j];
  if (p.gamma) ln_normalize(row, nv, lane, p.D, p.eps);     // gamma == NULL: plain cat + pos (DINO ViT has no ln_pre)
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    const int c = lane + 64 * j;
    if (c < nv) ((f32x4*)(p.out + r * p.D))[c] = p.gamma ? row.v[j] * ((const f32x4*)p.gamma)[c] + ((const f32x4*)p.beta)[c] : row.v[j];
  }
}

extern "C" int zh_assemble_tokens_ln(const float* patch_emb, const float* class_embedding, const float* pos_embed,
                                     const float* gamma, const float* beta, float eps, float* out,
                                     int B, int T, int D, hipStream_t stream) {
  ZH_CHECK_ARG(patch_emb && class_embedding && pos_embed && out, "zh_assemble_tokens_ln: null pointer");
  ZH_CHECK_ARG((gamma == nullptr) == (beta == nullptr), "zh_assemble_tokens_ln: gamma and beta must both be given or both null");
  ZH_CHECK_ARG(B > 0 && T > 1 && D % 4 == 0 && D <= 256 * LN_MAXV, "zh_assemble_tokens_ln: bad shape B=%d T=%d D=%d", B, T, D);
  AsmArgs p{patch_emb, class_embedding, pos_embed, gamma, beta, out, B, T, D, eps};
  hipLaunchKernelGGL(assemble_ln_kernel, dim3(zh_cdiv((long)B * T, 4)), dim3(256), 0, stream, p);
  ZH_CHECK_LAUNCH("zh_assemble_tokens_ln");
  return ZH_OK;
}

// ---- row L2 normalise: queries / ||queries||  (networks/zutis.py:515, no eps) -> fp16 and/or fp32
// f16_scale (a power of two; here and in the other producers of unit-norm rows): the fp16 / split-pair copy is stored as y * f16_scale and
// its consumer multiplies the finished accumulator by 1 / f16_scale (Act.out_scale).  A unit-norm row of 512 - 768 elements has
// |y| ~ 0.04: its lo half, ~2^-11 of that, is a SUBNORMAL fp16 number (below 6.1e-5: 19 significant bits for the pair instead of 22);
// times 2^10 both halves are normal numbers and nothing can overflow (|y| <= 1).
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float* x, float* out_f32, half_t* out_f16, int rows, int D, float eps,
                                                          long lo_plane, float f16_scale) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int nv = D >> 2;
  const f32x4* xp = (const f32x4*)(x + r * D);
  f32x4 v[LN_MAXV];
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j)
    if (lane + 64 * j < nv) {
      v[j] = xp[lane + 64 * j];
      q += (v[j][0] * v[j][0] + v[j][1] * v[j][1]) + (v[j][2] * v[j][2] + v[j][3] * v[j][3]);
    }
  const float inv = 1.0f / (sqrtf(wave_sum(q)) + eps);
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    const int c = lane + 64 * j;
    if (c < nv) {
      f32x4 y = v[j] * inv;
      if (out_f32) ((f32x4*)(out_f32 + r * D))[c] = y;
      if (out_f16) zh_store_h4(out_f16 + r * D + 4 * c, lo_plane, y * f16_scale);
    }
  }
}

extern "C" int zh_l2norm_rows(const float* x, float* out_f32, void* out_f16, float eps, int rows, int D, long lo_plane, float f16_scale,
                              hipStream_t stream) {
  ZH_CHECK_ARG(x && (out_f32 || out_f16) && rows > 0 && lo_plane % 4 == 0 && f16_scale > 0.f, "zh_l2norm_rows: bad arguments");
  ZH_CHECK_ARG(D % 4 == 0 && D <= 256 * LN_MAXV && D > 0, "zh_l2norm_rows: D=%d unsupported", D);
  hipLaunchKernelGGL(l2norm_rows_kernel, dim3(zh_cdiv(rows, 4)), dim3(256), 0, stream, x, out_f32, (half_t*)out_f16, rows, D, eps,
                     lo_plane, f16_scale);
  ZH_CHECK_LAUNCH("zh_l2norm_rows");
  return ZH_OK;
}

// ---- per-image LayerNorm over the whole (h,w,c) volume, no affine, then per-pixel L2 normalise
//      networks/zutis.py:320-322:  F.layer_norm(x, x.shape[1:]);  x / (||x||_c + 1e-7)
// Pass 1: per-chunk (count, mean, M2) partials (deterministic, no atomics).  Pass 2: every wave combines the
// image's partials with Chan's formula in fp64, then normalises one pixel row.
#define GLN_CHUNK 4096  // floats per partial block (256 threads x 4 float4)

__global__ __launch_bounds__(256) void gln_partial_kernel(const float* x, float* part, long per_image, int nchunks) {
  const int img = blockIdx.y, ch = blockIdx.x;
  const long base = (long)ch * GLN_CHUNK;
  const float* xp = x + (long)img * per_image + base;
  const long n = per_image - base < GLN_CHUNK ? per_image - base : GLN_CHUNK;  // multiple of 4
  f32x4 v[4];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long e = ((long)threadIdx.x + 256 * j) * 4;
    if (e < n) {
      v[j] = *(const f32x4*)(xp + e);
      s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    }
  }
  __shared__ float red[4];
  __shared__ float bc;
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) bc = ((red[0] + red[1]) + (red[2] + red[3])) / (float)n;
  __syncthreads();
  const float mean = bc;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long e = ((long)threadIdx.x + 256 * j) * 4;
    if (e < n) {
      f32x4 d = v[j] - mean;
      q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
    }
  }
  q = wave_sum(q);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q;
  __syncthreads();
  if (threadIdx.x == 0) {
    float* o = part + ((long)img * nchunks + ch) * 4;
    o[0] = (float)n; o[1] = mean; o[2] = (red[0] + red[1]) + (red[2] + red[3]); o[3] = 0.f;
  }
}

__global__ __launch_bounds__(256) void gln_apply_kernel(const float* x, const float* part, float* out_f32, half_t* out_f16,
                                                        long per_image, int nchunks, int M, int C, float eps, float l2_eps,
                                                        long lo_plane, float f16_scale, int* status) {
  const int img = blockIdx.y;
  const int lane = threadIdx.x & 63;
  // combine partials (every wave redundantly; nchunks is a few hundred)
  const float* pp = part + (long)img * nchunks * 4;
  double sn = 0.0, sm = 0.0;
  for (int i = lane; i < nchunks; i += 64) { sn += pp[4 * i]; sm += (double)pp[4 * i] * pp[4 * i + 1]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { sn += __shfl_xor(sn, o, 64); sm += __shfl_xor(sm, o, 64); }
  const double mean = sm / sn;
  double m2 = 0.0;
  for (int i = lane; i < nchunks; i += 64) {
    const double d = (double)pp[4 * i + 1] - mean;
    m2 += (double)pp[4 * i + 2] + (double)pp[4 * i] * d * d;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m2 += __shfl_xor(m2, o, 64);
  const float fmean = (float)mean;
  const float rstd = (float)(1.0 / sqrt(m2 / sn + (double)eps));
  if (status && blockIdx.x == 0 && threadIdx.x == 0 && !(m2 / sn < (double)INFINITY)) atomicOr(status, ZH_STATUS_NONFINITE);

  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= M) return;
  const int nv = C >> 2;
  const f32x4* xp = (const f32x4*)(x + (long)img * per_image + r * C);
  f32x4 v[LN_MAXV];
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j)
    if (lane + 64 * j < nv) {
      v[j] = (xp[lane + 64 * j] - fmean) * rstd;
      q += (v[j][0] * v[j][0] + v[j][1] * v[j][1]) + (v[j][2] * v[j][2] + v[j][3] * v[j][3]);
    }
  const float inv = 1.0f / (sqrtf(wave_sum(q)) + l2_eps);
  const long ob = (long)img * per_image + r * C;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    const int c = lane + 64 * j;
    if (c < nv) {
      f32x4 y = v[j] * inv;
      if (out_f32) ((f32x4*)(out_f32 + ob))[c] = y;
      if (out_f16) zh_store_h4(out_f16 + ob + 4 * c, lo_plane, y * f16_scale);
    }
  }
}

extern "C" size_t zh_global_ln_l2_workspace_size(int B, int M, int C) {
  const long per_image = (long)M * C;
  return (size_t)B * zh_cdiv(per_image, GLN_CHUNK) * 4 * sizeof(float);
}

extern "C" int zh_global_ln_l2(const float* x, float* out_f32, void* out_f16, float eps, float l2_eps,
                               int B, int M, int C, void* workspace, size_t workspace_bytes, long lo_plane, float f16_scale, int* status,
                               hipStream_t stream) {
  ZH_CHECK_ARG(x && (out_f32 || out_f16) && B > 0 && M > 0 && lo_plane % 4 == 0 && f16_scale > 0.f, "zh_global_ln_l2: bad arguments");
  ZH_CHECK_ARG(C % 4 == 0 && C <= 256 * LN_MAXV && C > 0, "zh_global_ln_l2: C=%d unsupported", C);
  ZH_CHECK_ARG(B < 65536, "zh_global_ln_l2: batch too large");
  const long per_image = (long)M * C;
  const int nchunks = zh_cdiv(per_image, GLN_CHUNK);
  if (workspace_bytes < zh_global_ln_l2_workspace_size(B, M, C) || !workspace) {
    zh_set_error("zh_global_ln_l2: workspace too small (%zu < %zu)", workspace_bytes, zh_global_ln_l2_workspace_size(B, M, C));
    return ZH_ERR_WORKSPACE;
  }
  hipLaunchKernelGGL(gln_partial_kernel, dim3(nchunks, B), dim3(256), 0, stream, x, (float*)workspace, per_image, nchunks);
  hipLaunchKernelGGL(gln_apply_kernel, dim3(zh_cdiv(M, 4), B), dim3(256), 0, stream, x, (const float*)workspace, out_f32,
                     (half_t*)out_f16, per_image, nchunks, M, C, eps, l2_eps, lo_plane, f16_scale, status);
  ZH_CHECK_LAUNCH("zh_global_ln_l2");
  return ZH_OK;
}
