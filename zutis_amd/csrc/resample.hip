// Resampling / layout kernels: im2col, bicubic pos-embed, x2 bilinear (channels-last), sine PE,
// fused bilinear-upsample + argmax, bilinear NCHW, casts (gfx950).  All HBM/L2-bound byte movers.
//
// Arithmetic order of the bilinear kernels follows ATen's CPU kernels exactly (explicit __fmaf_rn /
// __fmul_rn, see oracle/resample.py) so the fused argmax is bit-identical to
// torch.argmax(F.interpolate(..., mode="bilinear"), dim=1) on the same low-res logits, ties included.
#include "common.h"

// ---- im2col for the stride==kernel patch-embed conv (networks/clip_arch.py:340,378): pure re-index.
//      out[(b,py,px)][c*p*p + i*p + j] = x[b,c,py*p+i,px*p+j], zero padded to Kpad columns, fp16.
__global__ __launch_bounds__(128) void im2col_kernel(const float* x, half_t* out, int B, int Cin, int H, int W, int p,
                                                     int gh, int gw, int Kpad, int vec, long lo_plane) {
  // one workgroup per patch (= GEMM row): the (b, py, px) split is block-uniform; a thread converts 8 consecutive k =
  // 8 consecutive pixels of one patch row (two 16-byte loads, one 16-byte store) when p % 8 == 0 and rows are 16-B aligned
  const int row = blockIdx.x;
  const int px = row % gw;
  const int t = row / gw;
  const int py = t % gh, b = t / gh;
  const int pp = p * p, kreal = Cin * pp;
  half_t* orow = out + (long)row * Kpad;
  for (int k0 = threadIdx.x * 8; k0 < Kpad; k0 += 128 * 8) {
    half8_t o;
    if (vec && k0 + 8 <= kreal) {
      const int c = k0 / pp, ij = k0 - c * pp, i = ij / p, j = ij - i * p;
      const int yy = py * p + i, xx = px * p + j;
      if (yy < H && xx + 8 <= W) {
        const f32x4* src = (const f32x4*)(x + (((long)b * Cin + c) * H + yy) * W + xx);
        const f32x4 v0 = src[0], v1 = src[1];
        zh_store_h4(orow + k0, lo_plane, v0);
        zh_store_h4(orow + k0 + 4, lo_plane, v1);
        continue;
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = k0 + e;
      float v = 0.f;
      if (k < kreal) {
        const int c = k / pp, ij = k - c * pp, i = ij / p, j = ij - i * p;
        const int yy = py * p + i, xx = px * p + j;     // beyond the image only with pad_to_patch (zero padding,
        if (yy < H && xx < W) v = x[(((long)b * Cin + c) * H + yy) * W + xx];   // selfmask/vision_transformer.py:260-267)
      }
      half_t hv = (half_t)v;
      asm volatile("" : "+v"(hv));                                  // one conversion only: see zh_store_h4 (common.h)
      o[e] = hv;
      if (lo_plane) orow[lo_plane + k] = (half_t)(v - (float)hv);
    }
    *(half8_t*)(orow + k0) = o;
  }
}

extern "C" int zh_im2col_f16(const float* x, void* out, int B, int Cin, int H, int W, int patch, int Kpad, int pad_to_patch,
                             long lo_plane, hipStream_t stream) {
  ZH_CHECK_ARG(lo_plane % 8 == 0, "zh_im2col_f16: lo_plane must be a multiple of 8 halves");
  ZH_CHECK_ARG(x && out && B > 0 && Cin > 0 && patch > 0 && H > 0 && W > 0, "zh_im2col_f16: bad arguments");
  ZH_CHECK_ARG(pad_to_patch || (H >= patch && W >= patch), "zh_im2col_f16: image smaller than one patch");
  ZH_CHECK_ARG(Kpad >= Cin * patch * patch, "zh_im2col_f16: Kpad too small");
  const int gh = pad_to_patch ? (H + patch - 1) / patch : (H - patch) / patch + 1;
  const int gw = pad_to_patch ? (W + patch - 1) / patch : (W - patch) / patch + 1;
  ZH_CHECK_ARG(Kpad % 8 == 0 && ((uintptr_t)out & 15) == 0, "zh_im2col_f16: Kpad must be a multiple of 8 and out 16-byte aligned");
  const long rows = (long)B * gh * gw;
  ZH_CHECK_ARG(rows < (1L << 31), "zh_im2col_f16: too many patches");
  const int vec = (patch % 8 == 0) && (W % 4 == 0) && (((uintptr_t)x & 15) == 0);
  hipLaunchKernelGGL(im2col_kernel, dim3((unsigned)rows), dim3(128), 0, stream, x, (half_t*)out, B, Cin, H, W, patch, gh, gw, Kpad, vec,
                     lo_plane);
  ZH_CHECK_LAUNCH("zh_im2col_f16");
  return ZH_OK;
}

// ---- bicubic positional-embedding resample (networks/clip_arch.py:356-374; selfmask/vision_transformer.py:377-401)
//      pos [1+g*g, D] -> out [1+h*w, D]; row 0 (cls) copied.  Keys cubic A=-0.75, align_corners=False,
//      src = fma(scale, dst+0.5, -0.5) (no clamp), taps clamped to [0,g-1].  `scale_*` is the float32 coordinate
//      scale: float(1/scale_factor) for the CLIP form (g/(h+0.1), NOT g/h), in/out for the size= form.
__device__ __forceinline__ void cubic_coeffs(float t, float w[4]) {
  const float A = -0.75f;
  const float x0 = t + 1.0f, x3 = 2.0f - t, x2 = 1.0f - t;
  w[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
  w[1] = ((A + 2.0f) * t - (A + 3.0f)) * t * t + 1.0f;
  w[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
  w[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
}

__global__ __launch_bounds__(256) void posembed_bicubic_kernel(const float* pos, float* out, int g, int h, int w, int D,
                                                               float scale_h, float scale_w, int has_cls) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)(h * w + has_cls) * D;
  if (idx >= total) return;
  const int d = (int)(idx % D);
  const int t = (int)(idx / D);
  if (has_cls && t == 0) { out[idx] = pos[d]; return; }
  const int oy = (t - has_cls) / w, ox = (t - has_cls) - oy * w;
  const float sy = __fmaf_rn(scale_h, (float)oy + 0.5f, -0.5f), sx = __fmaf_rn(scale_w, (float)ox + 0.5f, -0.5f);
  const float fy = floorf(sy), fx = floorf(sx);
  float wy[4], wx[4];
  cubic_coeffs(sy - fy, wy);
  cubic_coeffs(sx - fx, wx);
  const int iy = (int)fy, ix = (int)fx;
  const float* base = pos + (long)has_cls * D + d;
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int yy = min(max(iy - 1 + a, 0), g - 1);
    float r = 0.f;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int xx = min(max(ix - 1 + b, 0), g - 1);
      const float v = base[(long)(yy * g + xx) * D];
      r = b == 0 ? __fmul_rn(v, wx[0]) : __fmaf_rn(v, wx[b], r);
    }
    acc = a == 0 ? __fmul_rn(r, wy[0]) : __fmaf_rn(r, wy[a], acc);
  }
  out[idx] = acc;
}

extern "C" int zh_posembed_bicubic(const float* pos, float* out, int grid, int h, int w, int D, float scale_h, float scale_w,
                                   int has_cls, hipStream_t stream) {
  ZH_CHECK_ARG(pos && out && grid > 0 && h > 0 && w > 0 && D > 0, "zh_posembed_bicubic: bad arguments");
  const long total = (long)(h * w + (has_cls ? 1 : 0)) * D;
  hipLaunchKernelGGL(posembed_bicubic_kernel, dim3(zh_cdiv(total, 256)), dim3(256), 0, stream, pos, out, grid, h, w, D, scale_h, scale_w, has_cls ? 1 : 0);
  ZH_CHECK_LAUNCH("zh_posembed_bicubic");
  return ZH_OK;
}

// ---- x2 bilinear upsample, channels-last tokens (networks/zutis.py:491-495): [B,h,w,D] fp32 -> [B,2h,2w,D]
//      fp16 and/or fp32.  src = max(0.5*(dst+0.5)-0.5, 0); weights {0.25,0.75} (edges clamp).
__global__ __launch_bounds__(256) void upsample2x_cl_kernel(const float* x, float* out_f32, half_t* out_f16, int B, int h, int w, int D,
                                                          long lo_plane, int relu) {
  // One workgroup per 2x2 output quad {2j+1,2j+2} x {2k+1,2k+2}, j in [-1,h-1], k in [-1,w-1]: the four outputs interpolate the
  // same four inputs (rows j,j+1 x cols k,k+1, clamped), so every input float4 is loaded once per quad instead of once per
  // output (the per-output form re-read 4x the tensor through L2 and ran at 2 TB/s).  Each output still evaluates ATen's own
  // source-index / weight formulas, so results are bit-identical to the per-output kernel.
  const int nv = D >> 2;
  const int qw = w + 1, qh = h + 1;
  const int quad = blockIdx.x;
  const int k = quad % qw - 1;
  const int t = quad / qw;
  const int j = t % qh - 1;
  const int b = t / qh;
  const int ya = max(j, 0), yb = min(j + 1, h - 1), xa = max(k, 0), xc = min(k + 1, w - 1);
  const f32x4* xb = (const f32x4*)(x + (long)b * h * w * D);
  const f32x4* p00 = xb + (long)(ya * w + xa) * nv;
  const f32x4* p01 = xb + (long)(ya * w + xc) * nv;
  const f32x4* p10 = xb + (long)(yb * w + xa) * nv;
  const f32x4* p11 = xb + (long)(yb * w + xc) * nv;
  float ly1[2], lx1[2];
  long orow[2];
  int ocol[2];
  bool vy[2], vx[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int oy = 2 * j + 1 + s, ox = 2 * k + 1 + s;
    vy[s] = oy >= 0 && oy < 2 * h;
    vx[s] = ox >= 0 && ox < 2 * w;
    const float sy = fmaxf(__fmaf_rn(0.5f, (float)oy + 0.5f, -0.5f), 0.f), sx = fmaxf(__fmaf_rn(0.5f, (float)ox + 0.5f, -0.5f), 0.f);
    const int y0 = min((int)sy, h - 1), x0 = min((int)sx, w - 1);
    ly1[s] = fminf(fmaxf(sy - (float)y0, 0.f), 1.f);
    lx1[s] = fminf(fmaxf(sx - (float)x0, 0.f), 1.f);
    orow[s] = ((long)b * 2 * h + oy) * (2 * w);
    ocol[s] = ox;
  }
  for (int c = threadIdx.x; c < nv; c += blockDim.x) {
    const f32x4 v00 = p00[c], v01 = p01[c], v10 = p10[c], v11 = p11[c];
#pragma unroll
    for (int sy = 0; sy < 2; ++sy)
#pragma unroll
      for (int sx = 0; sx < 2; ++sx) {
        if (!(vy[sy] && vx[sx])) continue;
        const float lx0 = 1.f - lx1[sx], ly0 = 1.f - ly1[sy];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float r0 = __fmaf_rn(v00[e], lx0, __fmul_rn(v01[e], lx1[sx]));
          const float r1 = __fmaf_rn(v10[e], lx0, __fmul_rn(v11[e], lx1[sx]));
          o[e] = __fmaf_rn(r0, ly0, __fmul_rn(r1, ly1[sy]));
          if (relu) o[e] = fmaxf(o[e], 0.0f);
        }
        const long oi = (orow[sy] + ocol[sx]) * nv + c;
        if (out_f32) ((f32x4*)out_f32)[oi] = o;
        if (out_f16) zh_store_h4(out_f16 + 4 * oi, lo_plane, o);
      }
  }
}

extern "C" int zh_upsample2x_bilinear_cl(const float* x, float* out_f32, void* out_f16, int B, int h, int w, int D, long lo_plane,
                                         int relu, hipStream_t stream) {
  ZH_CHECK_ARG(x && (out_f32 || out_f16) && B > 0 && h > 0 && w > 0 && D > 0 && D % 4 == 0 && lo_plane % 4 == 0,
               "zh_upsample2x_bilinear_cl: bad arguments");
  const long quads = (long)B * (h + 1) * (w + 1);          // one workgroup per 2x2 output quad
  ZH_CHECK_ARG(quads < (1L << 31), "zh_upsample2x_bilinear_cl: too many output quads");
  const int nv = D / 4;
  const int threads = nv >= 256 ? 256 : ((nv + 63) / 64) * 64;
  hipLaunchKernelGGL(upsample2x_cl_kernel, dim3((unsigned)quads), dim3(threads), 0, stream, x, out_f32, (half_t*)out_f16, B, h, w, D,
                     lo_plane, relu);
  ZH_CHECK_LAUNCH("zh_upsample2x_bilinear_cl");
  return ZH_OK;
}

// ---- sine positional embedding (networks/positional_embedding.py:29-52, normalize=True, scale=2*pi)
//      out[(y*w+x)][c], c<D/2: y part, else x part; even c sin, odd c cos; dim_t = T^(2*(c/2)/(D/2)).
__global__ __launch_bounds__(256) void sine_pe_kernel(float* out, int h, int w, int D, float temperature) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)h * w * D;
  if (idx >= total) return;
  const int c = (int)(idx % D);
  const int t = (int)(idx / D);
  const int y = t / w, x = t - y * w;
  const int npf = D >> 1;
  const int i = c < npf ? c : c - npf;
  const float e = c < npf ? (float)(y + 1) / ((float)h + 1e-6f) * 6.283185307179586f
                          : (float)(x + 1) / ((float)w + 1e-6f) * 6.283185307179586f;
  const float dim_t = powf(temperature, (float)(2 * (i / 2)) / (float)npf);
  const float a = e / dim_t;
  out[idx] = (i & 1) ? cosf(a) : sinf(a);
}

extern "C" int zh_sine_pe(float* out, int h, int w, int D, float temperature, hipStream_t stream) {
  ZH_CHECK_ARG(out && h > 0 && w > 0 && D > 0 && D % 4 == 0, "zh_sine_pe: bad arguments");
  hipLaunchKernelGGL(sine_pe_kernel, dim3(zh_cdiv((long)h * w * D, 256)), dim3(256), 0, stream, out, h, w, D, temperature);
  ZH_CHECK_LAUNCH("zh_sine_pe");
  return ZH_OK;
}

// ---- out_f16[r][:] = fp16(a_f16[r][:] + add_f32[r % add_rows][:])   (memory + pos, networks/transformer.py:281)
// Split pairs (a_lo_plane / lo_plane != 0): the sum is formed from hi + lo in fp32 and re-split.
__global__ __launch_bounds__(256) void add_rowperiodic_kernel(const half_t* a, const float* add, half_t* out, long rows, int D, int add_rows,
                                                              long a_lo_plane, long lo_plane) {
  const int nv = D >> 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * nv) return;
  const long r = idx / nv;
  const int c = (int)(idx - r * nv);
  const half4_t h = ((const half4_t*)a)[idx];
  const f32x4 p = ((const f32x4*)add)[(r % add_rows) * nv + c];
  f32x4 v = {(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
  if (a_lo_plane) {
    const half4_t l = *(const half4_t*)(a + a_lo_plane + 4 * idx);
    v += (f32x4){(float)l[0], (float)l[1], (float)l[2], (float)l[3]};
  }
  zh_store_h4(out + 4 * idx, lo_plane, v + p);
}

extern "C" int zh_add_rowperiodic_f16(const void* a, const float* add, void* out, long rows, int D, int add_rows, long a_lo_plane,
                                      long lo_plane, hipStream_t stream) {
  ZH_CHECK_ARG(a && add && out && rows > 0 && D > 0 && D % 4 == 0 && add_rows > 0 && a_lo_plane % 4 == 0 && lo_plane % 4 == 0,
               "zh_add_rowperiodic_f16: bad arguments");
  hipLaunchKernelGGL(add_rowperiodic_kernel, dim3(zh_cdiv(rows * (D / 4), 256)), dim3(256), 0, stream, (const half_t*)a, add, (half_t*)out, rows, D, add_rows,
                     a_lo_plane, lo_plane);
  ZH_CHECK_LAUNCH("zh_add_rowperiodic_f16");
  return ZH_OK;
}

// ---- fp32 -> fp16 cast (optionally adding a row-periodic fp32 matrix first)
__global__ __launch_bounds__(256) void cast_kernel(const float* x, const float* add, half_t* out, long n4, int nv, int add_rows,
                                                   long lo_plane, float f16_scale) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n4) return;
  f32x4 v = ((const f32x4*)x)[idx];
  if (add) {
    const long r = idx / nv;
    v += ((const f32x4*)add)[(r % add_rows) * nv + (idx - r * nv)];
  }
  zh_store_h4(out + 4 * idx, lo_plane, v * f16_scale);          // f16_scale: see l2norm_rows_kernel (norm.hip)
}

extern "C" int zh_cast_f32_f16(const float* x, const float* add, int add_rows, void* out, long rows, int D, long lo_plane, float f16_scale,
                               hipStream_t stream) {
  ZH_CHECK_ARG(x && out && rows > 0 && D > 0 && D % 4 == 0 && lo_plane % 4 == 0 && f16_scale > 0.f, "zh_cast_f32_f16: bad arguments");
  ZH_CHECK_ARG(!add || add_rows > 0, "zh_cast_f32_f16: add needs add_rows");
  const long n4 = rows * (D / 4);
  hipLaunchKernelGGL(cast_kernel, dim3(zh_cdiv(n4, 256)), dim3(256), 0, stream, x, add, (half_t*)out, n4, D / 4, add_rows > 0 ? add_rows : 1,
                     lo_plane, f16_scale);
  ZH_CHECK_LAUNCH("zh_cast_f32_f16");
  return ZH_OK;
}

// ---- bilinear index/weight, ATen compute_source_index_and_lambda (align_corners=False)
struct LinW { int i0, i1; float l0, l1; };
__device__ __forceinline__ LinW lin_weights(int dst, int in_size, int out_size, float scale) {
  LinW r;
  if (in_size == out_size) { r.i0 = r.i1 = dst; r.l0 = 1.f; r.l1 = 0.f; return r; }
  const float src = fmaxf(__fmaf_rn(scale, (float)dst + 0.5f, -0.5f), 0.f);
  r.i0 = min((int)src, in_size - 1);
  r.i1 = min(r.i0 + 1, in_size - 1);
  r.l1 = fminf(fmaxf(src - (float)r.i0, 0.f), 1.f);
  r.l0 = 1.f - r.l1;
  return r;
}

// ---- fused bilinear upsample + argmax over classes (networks/zutis.py:366-372), never materialising
//      [B,n,H,W].  logits [B,n,h,w] fp32 (NCHW, low-res) -> labels int64 [B,H,W]; first index on ties.
__global__ __launch_bounds__(256) void upsample_argmax_kernel(const float* lo, long long* labels, int B, int n, int h, int w,
                                                              int H, int W, float scale_h, float scale_w) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)B * H * W;
  if (idx >= total) return;
  const int ox = (int)(idx % W);
  const long t = idx / W;
  const int oy = (int)(t % H), b = (int)(t / H);
  const LinW wy = lin_weights(oy, h, H, scale_h), wx = lin_weights(ox, w, W, scale_w);
  const float* p = lo + (long)b * n * h * w;
  const int o00 = wy.i0 * w + wx.i0, o01 = wy.i0 * w + wx.i1, o10 = wy.i1 * w + wx.i0, o11 = wy.i1 * w + wx.i1;
  float best = 0.f;
  int besti = 0;
  for (int c = 0; c < n; ++c, p += (long)h * w) {
    const float r0 = __fmaf_rn(p[o00], wx.l0, __fmul_rn(p[o01], wx.l1));
    const float r1 = __fmaf_rn(p[o10], wx.l0, __fmul_rn(p[o11], wx.l1));
    const float v = __fmaf_rn(r0, wy.l0, __fmul_rn(r1, wy.l1));
    // torch.argmax: first maximal index; NaN is treated as maximal (propagates)
    if (c == 0 || v > best || (v != v && best == best)) { best = v; besti = c; }
  }
  labels[idx] = besti;
}

// LDS two-phase variant: a block owns a 32 x 32 output tile (each thread 4 pixels of one column) and walks the classes in
// chunks of UA_CH.  Per chunk: (0) stage the low-res window the tile touches (7 x 7 pixels at the 8x upsampling of 336 px),
// (1) interpolate every window row ONCE along x for the tile's 32 columns — R[c][row][ox] = fma(v0, lx0, v1*lx1), the value
// every output row of the tile would otherwise recompute — (2) per pixel and class: two conflict-free LDS reads, one
// fma(r0, ly0, r1*ly1), compare/select.  Same operations in the same order as ATen => bit-identical labels; 5 VALU + 2 LDS
// reads per (pixel, class) instead of 8 + 4.
#define UA_TH 32
#define UA_TW 32
#define UA_CH 32
#define UA_PX (UA_TH * UA_TW / 256)
__global__ __launch_bounds__(256) void upsample_argmax_lds_kernel(const float* lo, long long* labels, int n, int h, int w,
                                                                  int H, int W, float scale_h, float scale_w, int tiles_x, int tiles_y,
                                                                  int wr_max, int wc_max) {
  extern __shared__ __attribute__((aligned(16))) float ua_lds[];
  float* win = ua_lds;                                   // [UA_CH][wr][wc]
  float* R = ua_lds + UA_CH * wr_max * wc_max;           // [UA_CH][wr][UA_TW]
  const int b = blockIdx.x / (tiles_x * tiles_y);
  const int t = blockIdx.x - b * tiles_x * tiles_y;
  const int ty = t / tiles_x, tx = t - ty * tiles_x;
  const int oy0 = ty * UA_TH, ox0 = tx * UA_TW;
  const int oy1 = min(oy0 + UA_TH, H) - 1, ox1 = min(ox0 + UA_TW, W) - 1;
  const int y_lo = lin_weights(oy0, h, H, scale_h).i0, y_hi = lin_weights(oy1, h, H, scale_h).i1;
  const int x_lo = lin_weights(ox0, w, W, scale_w).i0, x_hi = lin_weights(ox1, w, W, scale_w).i1;
  const int wr = y_hi - y_lo + 1, wc = x_hi - x_lo + 1, wsz = wr * wc;
  const int txl = threadIdx.x % UA_TW, tyl = threadIdx.x / UA_TW;      // column in the tile, first row (rows tyl + 8 i)
  const int ox = min(ox0 + txl, W - 1);
  const LinW wx = lin_weights(ox, w, W, scale_w);
  const int xa = wx.i0 - x_lo, xb = wx.i1 - x_lo;
  int ra[UA_PX], rb[UA_PX];
  float l0[UA_PX], l1[UA_PX], best[UA_PX];
  int besti[UA_PX];
#pragma unroll
  for (int i = 0; i < UA_PX; ++i) {
    const int oy = min(oy0 + tyl + (256 / UA_TW) * i, H - 1);
    const LinW wy = lin_weights(oy, h, H, scale_h);
    ra[i] = (wy.i0 - y_lo) * (UA_CH * UA_TW) + txl;       // R is [row][class][column]: the class stride is a constant
    rb[i] = (wy.i1 - y_lo) * (UA_CH * UA_TW) + txl;
    l0[i] = wy.l0; l1[i] = wy.l1;
    best[i] = -INFINITY; besti[i] = 0;                    // all -inf -> index 0, as torch.argmax
  }
  __shared__ int nan_seen;
  const float* p = lo + (long)b * n * h * w;
  for (int c0 = 0; c0 < n; c0 += UA_CH) {
    const int nc = min(UA_CH, n - c0);
    __syncthreads();                                      // previous chunk's R / win no longer read
    if (threadIdx.x == 0) nan_seen = 0;
    // (0) window: wave per class, lane per window pixel — the (row, col) split of the pixel index is hoisted out of the class loop
    for (int r = threadIdx.x & 63; r < wsz; r += 64) {
      const int yy = r / wc, xx = r - yy * wc;
      const float* src = p + ((long)c0 * h + (y_lo + yy)) * w + (x_lo + xx);
      for (int c = threadIdx.x >> 6; c < nc; c += 4) win[c * wsz + r] = src[(long)c * h * w];
    }
    __syncthreads();
    // (1) x-interpolation of every window row for the tile's 32 columns: thread = (column txl, classes tyl, tyl+8, ...)
    bool nanflag = false;
    for (int c = tyl; c < nc; c += 256 / UA_TW) {
      const float* q = win + c * wsz;
      float* dst = R + c * UA_TW + txl;
      for (int yy = 0; yy < wr; ++yy, q += wc, dst += UA_CH * UA_TW) {
        const float r = __fmaf_rn(q[xa], wx.l0, __fmul_rn(q[xb], wx.l1));
        nanflag |= (r != r);
        *dst = r;
      }
    }
    if (nanflag) nan_seen = 1;
    __syncthreads();
    if (!nan_seen) {
      // fast path (no NaN anywhere in this chunk's window): strict > keeps the first maximum
#pragma unroll 4
      for (int c = 0; c < nc; ++c) {
#pragma unroll
        for (int i = 0; i < UA_PX; ++i) {
          const float v = __fmaf_rn(R[ra[i] + c * UA_TW], l0[i], __fmul_rn(R[rb[i] + c * UA_TW], l1[i]));
          const bool gt = v > best[i];                    // false forever once a NaN has won (best = NaN)
          best[i] = gt ? v : best[i];
          besti[i] = gt ? c0 + c : besti[i];
        }
      }
    } else {
      for (int c = 0; c < nc; ++c) {
#pragma unroll
        for (int i = 0; i < UA_PX; ++i) {
          const float v = __fmaf_rn(R[ra[i] + c * UA_TW], l0[i], __fmul_rn(R[rb[i] + c * UA_TW], l1[i]));
          // torch.argmax: first maximal index; NaN is treated as maximal (propagates)
          if (v > best[i] || (v != v && best[i] == best[i])) { best[i] = v; besti[i] = c0 + c; }
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < UA_PX; ++i) {
    const int oy = oy0 + tyl + (256 / UA_TW) * i;
    if (oy < H && ox0 + txl < W) labels[((long)b * H + oy) * W + ox0 + txl] = besti[i];
  }
}

// ---- round 6: the same tile walk with classes innermost in LDS, so that one ds_read_b128 brings FOUR classes of a pixel's row value
// and the two interpolation steps run as v_pk_mul_f32 / v_pk_fma_f32 on two classes at once (each half is the same IEEE multiply /
// fused multiply-add the scalar form does: bit-identical values), and the arg-max walks groups of four: two v_max3 give the group's
// maximum, ONE compare decides whether any of the four beats the running best (strict >: the first maximum stays), and only a wave in
// which some lane's best moved resolves WHICH of the four it was (first value equal to the group's maximum = torch's first-index
// rule).  Per (pixel, class): 0.5 packed arithmetic + 1 max / compare / select + 0.5 LDS b128 instead of 5 VALU + 2 ds_read_b32.
// What is class-independent (source rows / columns, the four weights) is computed once per tile as before.  The low-res window of
// the NEXT class chunk is fetched into registers while this chunk is interpolated and compared.
// profiles/NOTES.md round 6: 920 classes @518 px, 8 images: 830 us -> see there.
#define UA_CHP (UA_CH + 4)              // class stride in LDS (floats): 16-byte aligned, and lanes 36 floats apart spread over all 64 banks
#define UA_NPF 8                        // window values a thread carries for the next chunk (window <= 64 low-res pixels)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t ua_pk_mul(f32x2_t a, f32x2_t b) {
  f32x2_t r;
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f32x2_t ua_pk_fma(f32x2_t a, f32x2_t b, f32x2_t c) {
  f32x2_t r;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float ua_max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__global__ __launch_bounds__(256) void upsample_argmax_pk_kernel(const float* lo, long long* labels, int n, int h, int w,
                                                                 int H, int W, float scale_h, float scale_w, int tiles_x, int tiles_y,
                                                                 int wr_max, int wc_max) {
  extern __shared__ __attribute__((aligned(16))) float ua_lds[];
  float* win = ua_lds;                                   // [window pixel][UA_CHP]
  float* R = ua_lds + wr_max * wc_max * UA_CHP;          // [window row][UA_TW columns][UA_CHP]
  const int b = blockIdx.x / (tiles_x * tiles_y);
  const int t = blockIdx.x - b * tiles_x * tiles_y;
  const int ty = t / tiles_x, tx = t - ty * tiles_x;
  const int oy0 = ty * UA_TH, ox0 = tx * UA_TW;
  const int oy1 = min(oy0 + UA_TH, H) - 1, ox1 = min(ox0 + UA_TW, W) - 1;
  const int y_lo = lin_weights(oy0, h, H, scale_h).i0, y_hi = lin_weights(oy1, h, H, scale_h).i1;
  const int x_lo = lin_weights(ox0, w, W, scale_w).i0, x_hi = lin_weights(ox1, w, W, scale_w).i1;
  const int wr = y_hi - y_lo + 1, wc = x_hi - x_lo + 1, wsz = wr * wc;
  const int txl = threadIdx.x % UA_TW, tyl = threadIdx.x / UA_TW;      // stage 2: column in the tile, first row (rows tyl + 8 i); stage 1: column, class group
  const int ox = min(ox0 + txl, W - 1);
  const LinW wx = lin_weights(ox, w, W, scale_w);
  const int xa = (wx.i0 - x_lo) * UA_CHP + 4 * tyl, xb = (wx.i1 - x_lo) * UA_CHP + 4 * tyl;
  const f32x2_t wx0 = {wx.l0, wx.l0}, wx1 = {wx.l1, wx.l1};
  int ra[UA_PX], rb[UA_PX];
  f32x2_t l0[UA_PX], l1[UA_PX];
  float best[UA_PX];
  int besti[UA_PX];
#pragma unroll
  for (int i = 0; i < UA_PX; ++i) {
    const int oy = min(oy0 + tyl + (256 / UA_TW) * i, H - 1);
    const LinW wy = lin_weights(oy, h, H, scale_h);
    ra[i] = ((wy.i0 - y_lo) * UA_TW + txl) * UA_CHP;
    rb[i] = ((wy.i1 - y_lo) * UA_TW + txl) * UA_CHP;
    l0[i] = (f32x2_t){wy.l0, wy.l0}; l1[i] = (f32x2_t){wy.l1, wy.l1};
    best[i] = -INFINITY; besti[i] = 0;                    // all -inf -> index 0, as torch.argmax
  }
  __shared__ int nan_seen;
  const float* p = lo + (long)b * n * h * w;
  // the window of a chunk, pixel-fastest over the threads (a window row is contiguous in memory): value j of this thread is
  // (class e / wsz of the chunk, window pixel e % wsz), e = tid + 256 j
  int pf_src[UA_NPF], pf_dst[UA_NPF], pf_cls[UA_NPF];
#pragma unroll
  for (int j = 0; j < UA_NPF; ++j) {
    const int e = threadIdx.x + 256 * j;
    const int c = e / wsz, r = e - c * wsz;
    const int yy = r / wc, xx = r - yy * wc;
    pf_cls[j] = c < UA_CH ? c : (1 << 30);                // beyond the chunk: never fetched, never stored
    pf_src[j] = (c * h + y_lo + yy) * w + x_lo + xx;
    pf_dst[j] = r * UA_CHP + c;
  }
  float pf[UA_NPF];
  auto fetch = [&](int c0) {
    const float* q = p + (long)c0 * h * w;
#pragma unroll
    for (int j = 0; j < UA_NPF; ++j) pf[j] = c0 + pf_cls[j] < n ? q[pf_src[j]] : 0.f;
  };
  fetch(0);
  for (int c0 = 0; c0 < n; c0 += UA_CH) {
    __syncthreads();                                      // previous chunk's R / win no longer read
    if (threadIdx.x == 0) nan_seen = 0;
#pragma unroll
    for (int j = 0; j < UA_NPF; ++j)
      if (pf_cls[j] < UA_CH) win[pf_dst[j]] = pf[j];
    __syncthreads();
    if (c0 + UA_CH < n) fetch(c0 + UA_CH);                // in flight under stages 1 and 2
    // (1) x-interpolation of every window row for the tile's 32 columns: thread = (column txl, classes 4 tyl .. 4 tyl + 3)
    {
      bool nanflag = false;
      const int cls = c0 + 4 * tyl;
      float* dst = R + txl * UA_CHP + 4 * tyl;
      const float* q = win;
      for (int yy = 0; yy < wr; ++yy, q += wc * UA_CHP, dst += UA_TW * UA_CHP) {
        const f32x4 a = *(const f32x4*)(q + xa), bq = *(const f32x4*)(q + xb);
        const f32x2_t r01 = ua_pk_fma((f32x2_t){a[0], a[1]}, wx0, ua_pk_mul((f32x2_t){bq[0], bq[1]}, wx1));
        const f32x2_t r23 = ua_pk_fma((f32x2_t){a[2], a[3]}, wx0, ua_pk_mul((f32x2_t){bq[2], bq[3]}, wx1));
        f32x4 r = {r01[0], r01[1], r23[0], r23[1]};
        // classes past n (the last chunk): -inf, never a maximum, never a NaN
        r[0] = cls + 0 < n ? r[0] : -INFINITY; r[1] = cls + 1 < n ? r[1] : -INFINITY;
        r[2] = cls + 2 < n ? r[2] : -INFINITY; r[3] = cls + 3 < n ? r[3] : -INFINITY;
        nanflag |= (r[0] != r[0]) | (r[1] != r[1]) | (r[2] != r[2]) | (r[3] != r[3]);
        *(f32x4*)dst = r;
      }
      if (nanflag) nan_seen = 1;
    }
    __syncthreads();
    if (!nan_seen) {
      // fast path (no NaN anywhere in this chunk's window): strict > keeps the first maximum
#pragma unroll
      for (int g = 0; g < UA_CH / 4; ++g) {                // fully unrolled: the LDS offsets are immediates
        f32x4 a[UA_PX], bq[UA_PX];
#pragma unroll
        for (int i = 0; i < UA_PX; ++i) {                  // the group's eight reads together, then the arithmetic
          a[i] = *(const f32x4*)(R + ra[i] + 4 * g);
          bq[i] = *(const f32x4*)(R + rb[i] + 4 * g);
        }
        const int base = c0 + 4 * g;
#pragma unroll
        for (int i = 0; i < UA_PX; ++i) {
          const f32x2_t v01 = ua_pk_fma((f32x2_t){a[i][0], a[i][1]}, l0[i], ua_pk_mul((f32x2_t){bq[i][0], bq[i][1]}, l1[i]));
          const f32x2_t v23 = ua_pk_fma((f32x2_t){a[i][2], a[i][3]}, l0[i], ua_pk_mul((f32x2_t){bq[i][2], bq[i][3]}, l1[i]));
          const float m = ua_max3(ua_max3(v01[0], v01[1], v23[0]), v23[1], -INFINITY);
          const bool gt = m > best[i];                     // -inf padding / all -inf: never
          if (__any(gt)) {                                 // wave-uniform: which of the four it was matters only when some lane's best moved
            int k = base + 3;
            k = v23[0] == m ? base + 2 : k;
            k = v01[1] == m ? base + 1 : k;
            k = v01[0] == m ? base : k;                    // the FIRST value equal to the group's maximum
            besti[i] = gt ? k : besti[i];
            best[i] = gt ? m : best[i];
          }
        }
      }
    } else {
      const int nc = min(UA_CH, n - c0);
      for (int c = 0; c < nc; ++c) {
#pragma unroll
        for (int i = 0; i < UA_PX; ++i) {
          const float v = __fmaf_rn(R[ra[i] + c], l0[i][0], __fmul_rn(R[rb[i] + c], l1[i][0]));
          // torch.argmax: first maximal index; NaN is treated as maximal (propagates)
          if (v > best[i] || (v != v && best[i] == best[i])) { best[i] = v; besti[i] = c0 + c; }
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < UA_PX; ++i) {
    const int oy = oy0 + tyl + (256 / UA_TW) * i;
    if (oy < H && ox0 + txl < W) labels[((long)b * H + oy) * W + ox0 + txl] = besti[i];
  }
}

extern "C" int zh_upsample_argmax(const float* logits_lo, long long* labels, int B, int n, int h, int w, int H, int W,
                                  float scale_h, float scale_w, hipStream_t stream) {
  ZH_CHECK_ARG(logits_lo && labels && B > 0 && n > 0 && h > 0 && w > 0 && H > 0 && W > 0, "zh_upsample_argmax: bad arguments");
  // worst-case low-res window of a 32 x 32 output tile: ceil(extent * scale) + 2 rows / cols (identity sizes: extent)
  const int wr = (h == H ? UA_TH : (int)(UA_TH * scale_h) + 3), wc = (w == W ? UA_TW : (int)(UA_TW * scale_w) + 3);
  const size_t lds = (size_t)UA_CH * wr * (wc + UA_TW) * sizeof(float);
  const long tiles = (long)B * zh_cdiv(W, UA_TW) * zh_cdiv(H, UA_TH);
  const size_t lds_pk = (size_t)UA_CHP * wr * (wc + UA_TW) * sizeof(float);
  static const bool pk_off = [] { const char* e = getenv("ZH_UPSAMPLE_ARGMAX_PK"); return e && atoi(e) == 0; }();   // developer A/B: 0 = the round-2 kernel
  if (!pk_off && wr * wc <= 64 && lds_pk <= 48 * 1024 && tiles < (1L << 31)) {   // a window of <= 64 low-res pixels (upsampling by >= ~6x at 32 x 32 tiles)
    const int tiles_y = zh_cdiv(H, UA_TH), tiles_x = zh_cdiv(W, UA_TW);
    hipLaunchKernelGGL(upsample_argmax_pk_kernel, dim3((unsigned)tiles), dim3(256), lds_pk, stream, logits_lo, labels,
                       n, h, w, H, W, scale_h, scale_w, tiles_x, tiles_y, wr, wc);
  } else if (lds <= 48 * 1024 && tiles < (1L << 31)) {    // upsampling by >= ~2.5x; otherwise the direct kernel
    const int tiles_y = zh_cdiv(H, UA_TH), tiles_x = zh_cdiv(W, UA_TW);
    hipLaunchKernelGGL(upsample_argmax_lds_kernel, dim3((unsigned)tiles), dim3(256), lds, stream, logits_lo, labels,
                       n, h, w, H, W, scale_h, scale_w, tiles_x, tiles_y, wr, wc);
  } else {
    hipLaunchKernelGGL(upsample_argmax_kernel, dim3(zh_cdiv((long)B * H * W, 256)), dim3(256), 0, stream, logits_lo, labels, B, n, h, w, H, W, scale_h, scale_w);
  }
  ZH_CHECK_LAUNCH("zh_upsample_argmax");
  return ZH_OK;
}

// ---- bilinear NCHW upsample (return_logits path zutis.py:368-371 and instance masks zutis.py:422-423);
//      optional threshold output: mask_u8 = (value > threshold)
__global__ __launch_bounds__(256) void bilinear_nchw_kernel(const float* x, float* out, unsigned char* mask, float threshold,
                                                            long planes, int h, int w, int H, int W, float scale_h, float scale_w) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = planes * H * W;
  if (idx >= total) return;
  const int ox = (int)(idx % W);
  const long t = idx / W;
  const int oy = (int)(t % H);
  const long pl = t / H;
  const LinW wy = lin_weights(oy, h, H, scale_h), wx = lin_weights(ox, w, W, scale_w);
  const float* p = x + pl * h * w;
  const float r0 = __fmaf_rn(p[wy.i0 * w + wx.i0], wx.l0, __fmul_rn(p[wy.i0 * w + wx.i1], wx.l1));
  const float r1 = __fmaf_rn(p[wy.i1 * w + wx.i0], wx.l0, __fmul_rn(p[wy.i1 * w + wx.i1], wx.l1));
  const float v = __fmaf_rn(r0, wy.l0, __fmul_rn(r1, wy.l1));
  if (out) out[idx] = v;
  if (mask) mask[idx] = v > threshold ? 1 : 0;
}
// The thresholded-mask form at W % 4 == 0 (instance masks, zutis.py:422-423: 100 planes of 480 x 640 at batch 1): a block covers
// MASK_ROWS output rows of a plane — the column weights of a thread's 4 pixels are computed once for all of them, no 64-bit index divisions
// per pixel (the flat kernel above spent 74 us on them for 31 M pixels), 4 pixels leave as one 32-bit store; round 4: eight rows per block
// instead of one (48 000 blocks of 160 busy threads were most of the 36 us).  Same arithmetic per pixel, bit-identical masks.
#define MASK_ROWS 8
__global__ __launch_bounds__(256) void bilinear_mask_rows_kernel(const float* x, unsigned char* mask, float threshold, int h, int w, int H, int W,
                                                                 float scale_h, float scale_w, int row_groups) {
  const unsigned grp = blockIdx.x;                          // plane * row_groups + row group
  const unsigned pl = grp / (unsigned)row_groups, rg = grp - pl * (unsigned)row_groups;
  const int ox0 = (blockIdx.y * 256 + threadIdx.x) * 4;
  if (ox0 >= W) return;
  LinW wx[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) wx[k] = lin_weights(ox0 + k, w, W, scale_w);
  const float* px = x + (long)pl * h * w;
#pragma unroll
  for (int r = 0; r < MASK_ROWS; ++r) {
    const int oy = (int)rg * MASK_ROWS + r;
    if (oy >= H) break;
    const LinW wy = lin_weights(oy, h, H, scale_h);
    const float* p0 = px + (long)wy.i0 * w;
    const float* p1 = px + (long)wy.i1 * w;
    unsigned packed = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float r0 = __fmaf_rn(p0[wx[k].i0], wx[k].l0, __fmul_rn(p0[wx[k].i1], wx[k].l1));
      const float r1 = __fmaf_rn(p1[wx[k].i0], wx[k].l0, __fmul_rn(p1[wx[k].i1], wx[k].l1));
      const float v = __fmaf_rn(r0, wy.l0, __fmul_rn(r1, wy.l1));
      packed |= (v > threshold ? 1u : 0u) << (8 * k);
    }
    *(unsigned*)(mask + ((long)pl * H + oy) * W + ox0) = packed;
  }
}

// SelfMask inference tail (networks/selfmask/selfmask.py:207-221) without a host round trip: per image pick the query with
// the largest objectness logit (first maximum, NaN = maximum: np/torch argmax), bilinear-upsample ONLY that mask plane
// (ATen arithmetic as above), crop to [H,W] and threshold.
__global__ __launch_bounds__(256) void select_upsample_mask_kernel(const float* obj, const float* masks, unsigned char* out, long long* index,
                                                                   int Q, int h, int w, int H, int W, float scale_h, float scale_w, float threshold) {
  const int b = blockIdx.y;
  const float* o = obj + (long)b * Q;
  int best = 0;
  float bv = o[0];
  for (int q = 1; q < Q; ++q) {
    const float v = o[q];
    if (v > bv || (v != v && bv == bv)) { bv = v; best = q; }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) index[b] = best;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= H * W) return;
  const int ox = idx % W, oy = idx / W;
  const LinW wy = lin_weights(oy, h, H, scale_h), wx = lin_weights(ox, w, W, scale_w);
  const float* p = masks + ((long)b * Q + best) * h * w;
  const float r0 = __fmaf_rn(p[wy.i0 * w + wx.i0], wx.l0, __fmul_rn(p[wy.i0 * w + wx.i1], wx.l1));
  const float r1 = __fmaf_rn(p[wy.i1 * w + wx.i0], wx.l0, __fmul_rn(p[wy.i1 * w + wx.i1], wx.l1));
  const float v = __fmaf_rn(r0, wy.l0, __fmul_rn(r1, wy.l1));
  out[(long)b * H * W + idx] = v > threshold ? 1 : 0;
}

extern "C" int zh_select_upsample_mask(const float* objectness, const float* masks, unsigned char* out_u8, long long* index, int B, int Q,
                                       int h, int w, int H, int W, float scale_h, float scale_w, float threshold, hipStream_t stream) {
  ZH_CHECK_ARG(objectness && masks && out_u8 && index && B > 0 && Q > 0 && h > 0 && w > 0 && H > 0 && W > 0, "zh_select_upsample_mask: bad arguments");
  ZH_CHECK_ARG(B < 65536 && (long)H * W < (1L << 31), "zh_select_upsample_mask: shape exceeds grid limits");
  hipLaunchKernelGGL(select_upsample_mask_kernel, dim3(zh_cdiv((long)H * W, 256), B), dim3(256), 0, stream, objectness, masks, out_u8, index,
                     Q, h, w, H, W, scale_h, scale_w, threshold);
  ZH_CHECK_LAUNCH("zh_select_upsample_mask");
  return ZH_OK;
}

extern "C" int zh_upsample_bilinear_nchw(const float* x, float* out, unsigned char* mask_u8, float threshold, long planes,
                                         int h, int w, int H, int W, float scale_h, float scale_w, hipStream_t stream) {
  ZH_CHECK_ARG(x && (out || mask_u8) && planes > 0 && h > 0 && w > 0 && H > 0 && W > 0, "zh_upsample_bilinear_nchw: bad arguments");
  if (!out && W % 4 == 0 && ((uintptr_t)mask_u8 & 3) == 0 && planes * H < (1L << 31) && W <= 1024 * 65535) {
    const int row_groups = zh_cdiv(H, MASK_ROWS);
    hipLaunchKernelGGL(bilinear_mask_rows_kernel, dim3((unsigned)(planes * row_groups), zh_cdiv(W, 1024)), dim3(256), 0, stream, x, mask_u8, threshold,
                       h, w, H, W, scale_h, scale_w, row_groups);
    ZH_CHECK_LAUNCH("zh_upsample_bilinear_nchw");
    return ZH_OK;
  }
  hipLaunchKernelGGL(bilinear_nchw_kernel, dim3(zh_cdiv(planes * H * W, 256)), dim3(256), 0, stream, x, out, mask_u8, threshold, planes, h, w, H, W, scale_h, scale_w);
  ZH_CHECK_LAUNCH("zh_upsample_bilinear_nchw");
  return ZH_OK;
}

// ---- nearest-neighbour resize of a u8 mask (F.interpolate(mode="nearest"), datasets/index_dataset.py:215):
//      src = min(floor(dst * scale), in - 1), scale = float(in) / float(out) (ATen nearest_neighbor_compute_source_index)
__global__ __launch_bounds__(256) void resize_nearest_u8_kernel(const unsigned char* x, unsigned char* out, int h, int w, int H, int W,
                                                                float scale_h, float scale_w) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)H * W) return;
  const int oy = (int)(idx / W), ox = (int)(idx - (long)oy * W);
  const int sy = min((int)floorf(__fmul_rn((float)oy, scale_h)), h - 1);
  const int sx = min((int)floorf(__fmul_rn((float)ox, scale_w)), w - 1);
  out[idx] = x[(long)sy * w + sx];
}

extern "C" int zh_resize_nearest_u8(const unsigned char* x, unsigned char* out, int h, int w, int H, int W, float scale_h, float scale_w,
                                    hipStream_t stream) {
  ZH_CHECK_ARG(x && out && h > 0 && w > 0 && H > 0 && W > 0, "zh_resize_nearest_u8: bad arguments");
  hipLaunchKernelGGL(resize_nearest_u8_kernel, dim3(zh_cdiv((long)H * W, 256)), dim3(256), 0, stream, x, out, h, w, H, W, scale_h, scale_w);
  ZH_CHECK_LAUNCH("zh_resize_nearest_u8");
  return ZH_OK;
}

// ---- fill an f32 buffer (tgt = zeros, networks/zutis.py:164) as a kernel so it can live inside a launch plan
__global__ __launch_bounds__(256) void fill_f32_kernel(float* x, float v, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] = v;
}
extern "C" int zh_fill_f32(float* x, float value, long n, hipStream_t stream) {
  ZH_CHECK_ARG(x && n > 0, "zh_fill_f32: bad arguments");
  hipLaunchKernelGGL(fill_f32_kernel, dim3(zh_cdiv(n, 256)), dim3(256), 0, stream, x, value, n);
  ZH_CHECK_LAUNCH("zh_fill_f32");
  return ZH_OK;
}
