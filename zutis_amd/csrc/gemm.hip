// fp16-input / fp32-accumulate MFMA GEMM with fused epilogues (gfx950).
//
//   C[b][m][n] = act( sum_k A[b][m][k] * W[b][n][k] + bias[n] ) + R[b][m % res_rows][n]
//
// Both operands are K-contiguous ("NT" form = torch Linear layout, networks/clip_arch.py:304-310),
// so one kernel serves every contraction on the hot path: patch-embed conv-as-GEMM
// (clip_arch.py:378), QKV/out-proj/MLP (clip_arch.py:314-320), ffn1/ffn2 (zutis.py:546-549),
// decoder projections/FFN (transformer.py:272-290), the mask einsum (zutis.py:196-198, batched, sigmoid
// epilogue), the text-space projection (zutis.py:319) and the class-logit einsum (zutis.py:361-365).
//
// Design (MI355X): 128x128x64 block tile, 4 waves in 2x2, each wave 64x64 = 4x4 tiles of
// v_mfma_f32_16x16x32_f16.  Operand roles are swapped (MFMA-A = W rows, MFMA-B = A rows) so a lane's
// 4 accumulator registers are 4 consecutive n of one output row -> 16-byte row-major stores and
// float4 bias/residual loads.  Tiles are staged HBM->LDS with global_load_lds_dwordx4 (no VGPR round
// trip), double-buffered; the LDS image is lane-linear, bank conflicts are removed by XOR-ing the
// 16-byte chunk index with (row & 7) on the *source* address and on the ds_read_b128 address
// (conflict-free for the 16x16x32 operand maps).  Block ids are remapped so each XCD's L2 sees a
// contiguous range of tiles.
#include "common.h"

#define BM 128
#define BN 128
#define BK 64

struct GemmArgs {
  const half_t* A; long lda, sA;
  const half_t* W; long ldw, sW;
  void* C; long ldc, sC;
  const float* bias;
  const float* R; long ldr, sR; int res_rows;
  int M, N, K, act, nbm, nbn, vec_ok;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

__device__ __forceinline__ void stage_tile(const half_t* __restrict__ g, long ld, int row0, int nrows, int k0,
                                           half_t* s, int wave, int lane) {
  // tile = 128 rows x 64 halves (128 B per row); one wave-instruction fills 8 rows (1 KiB).
  const int rsub = lane >> 3;
  const int kc = (lane & 7) ^ rsub;  // logical 16-B chunk stored at physical chunk (lane&7) of row (..&7)==rsub
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int rg = wave * 4 + t;
    int row = row0 + rg * 8 + rsub;
    row = row < nrows ? row : nrows - 1;
    const half_t* src = g + (long)row * ld + k0 + kc * 8;
    __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(s + rg * 8 * BK), 16, 0, 0);
  }
}

template <int OUT_F16>
__global__ __launch_bounds__(256, 2) void gemm_f16_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) half_t smem[2 * (BM + BN) * BK];  // 64 KiB

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  // XCD-aware bijective remap: blocks b, b+8, ... share an XCD -> give each XCD a contiguous tile range
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tiles = p.nbm * p.nbn;
  const int batch = wg / tiles;
  const int trem = wg - batch * tiles;
  const int tm = trem / p.nbn, tn = trem - tm * p.nbn;
  const int m0 = tm * BM, n0 = tn * BN;

  const half_t* A = p.A + (long)batch * p.sA;
  const half_t* W = p.W + (long)batch * p.sW;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  stage_tile(A, p.lda, m0, p.M, 0, smem, wave, lane);
  stage_tile(W, p.ldw, n0, p.N, 0, smem + BM * BK, wave, lane);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();

  const int frow = lane & 15, fk = lane >> 4;
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) {
      half_t* nxt = smem + (cur ^ 1) * (BM + BN) * BK;
      stage_tile(A, p.lda, m0, p.M, (kt + 1) * BK, nxt, wave, lane);
      stage_tile(W, p.ldw, n0, p.N, (kt + 1) * BK, nxt + BM * BK, wave, lane);
    }
    const half_t* sA = smem + cur * (BM + BN) * BK;
    const half_t* sW = sA + BM * BK;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      half8_t fa[4], fw[4];
      const int kc = ks * 4 + fk;
      const int sw = (kc ^ (frow & 7)) * 8;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        fa[t] = *(const half8_t*)(sA + (wr * 64 + t * 16 + frow) * BK + sw);
        fw[t] = *(const half8_t*)(sW + (wc * 64 + t * 16 + frow) * BK + sw);
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[nt], fa[mt], acc[nt][mt], 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: lane owns rows m = ..+(lane&15), 4 consecutive n at 4*(lane>>4)
  const long cb = (long)batch * p.sC;
  const float* R = p.R ? p.R + (long)batch * p.sR : nullptr;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = m0 + wr * 64 + mt * 16 + frow;
    if (m >= p.M) continue;
    const long rrow = R ? (long)(m % p.res_rows) * p.ldr : 0;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = n0 + wc * 64 + nt * 16 + fk * 4;
      if (n >= p.N) continue;
      f32x4 v = acc[nt][mt];
      if (p.vec_ok) {
        if (p.bias) v += *(const f32x4*)(p.bias + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = zh_act(v[e], p.act);
        if (R) v += *(const f32x4*)(R + rrow + n);
        if (OUT_F16) {
          half4_t h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
          *(half4_t*)((half_t*)p.C + cb + (long)m * p.ldc + n) = h;
        } else {
          *(f32x4*)((float*)p.C + cb + (long)m * p.ldc + n) = v;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (n + e >= p.N) break;
          float x = v[e];
          if (p.bias) x += p.bias[n + e];
          x = zh_act(x, p.act);
          if (R) x += R[rrow + n + e];
          if (OUT_F16) ((half_t*)p.C)[cb + (long)m * p.ldc + n + e] = (half_t)x;
          else ((float*)p.C)[cb + (long)m * p.ldc + n + e] = x;
        }
      }
    }
  }
}

extern "C" int zh_gemm_f16(const void* A, long lda, long strideA, const void* W, long ldw, long strideW,
                           void* C, long ldc, long strideC, int out_f16,
                           const float* bias, const float* residual, long ldr, long strideR, int res_rows,
                           int act, int M, int N, int K, int batch, hipStream_t stream) {
  ZH_CHECK_ARG(A && W && C, "zh_gemm_f16: null operand");
  ZH_CHECK_ARG(M > 0 && N > 0 && K > 0 && batch > 0, "zh_gemm_f16: bad shape M=%d N=%d K=%d batch=%d", M, N, K, batch);
  ZH_CHECK_ARG(K % BK == 0, "zh_gemm_f16: K=%d must be a multiple of %d", K, BK);
  ZH_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && strideA % 8 == 0 && strideW % 8 == 0,
               "zh_gemm_f16: lda/ldw/strides must be multiples of 8 halves (16-byte rows)");
  ZH_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0, "zh_gemm_f16: A/W must be 16-byte aligned");
  ZH_CHECK_ARG(act >= 0 && act <= 4, "zh_gemm_f16: bad activation %d", act);
  ZH_CHECK_ARG(!residual || res_rows > 0, "zh_gemm_f16: residual needs res_rows > 0");
  GemmArgs p;
  p.A = (const half_t*)A; p.lda = lda; p.sA = strideA;
  p.W = (const half_t*)W; p.ldw = ldw; p.sW = strideW;
  p.C = C; p.ldc = ldc; p.sC = strideC;
  p.bias = bias; p.R = residual; p.ldr = ldr; p.sR = strideR; p.res_rows = res_rows;
  p.M = M; p.N = N; p.K = K; p.act = act;
  p.nbm = zh_cdiv(M, BM); p.nbn = zh_cdiv(N, BN);
  const int esz = out_f16 ? 2 : 4;
  p.vec_ok = (N % 4 == 0) && (ldc % 4 == 0) && (strideC % 4 == 0) && (((uintptr_t)C & (4 * esz - 1)) == 0) &&
             (!bias || ((uintptr_t)bias & 15) == 0) &&
             (!residual || (ldr % 4 == 0 && strideR % 4 == 0 && ((uintptr_t)residual & 15) == 0));
  const long nblk = (long)p.nbm * p.nbn * batch;
  ZH_CHECK_ARG(nblk < (1L << 31), "zh_gemm_f16: grid too large");
  if (out_f16) hipLaunchKernelGGL(gemm_f16_kernel<1>, dim3((unsigned)nblk), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(gemm_f16_kernel<0>, dim3((unsigned)nblk), dim3(256), 0, stream, p);
  ZH_CHECK_LAUNCH("zh_gemm_f16");
  return ZH_OK;
}
