// Few-row f16x3 GEMM (M <= a few hundred rows): the decoder's projections / FFN at batch 1 (Q = 100 query rows,
// networks/transformer.py:262-291), ffn2 (networks/zutis.py:514), SelfMask's 20-query decoder (networks/selfmask/selfmask.py).
//
//   C[b][m][n] = act( (sum_k A[b][m][k] * W[b][n][k]) * out_scale + bias[n] ) + R[b][m % res_rows][n]
//
// Why its own kernel: with 100 rows the LDS-ring kernel (gemm_kernel.h) runs 24 blocks of 64 x 64, each walking all of K
// slice by slice behind a barrier — 24 dependent load latencies, 13 - 28 us for 0.1 GFLOP (profiles/r03_c3_forward_*).
// Such a GEMM is a weight STREAM (2.4 - 6 MB read once) against a tiny activation block; what bounds it is latency, so every
// byte a block needs is requested at once:
//  * a block owns a 32 x 32 (MT x NT sub-tiles of 16 x 16) output tile; its four waves split K by interleaved 32-wide steps
//    (wave w takes steps w, w + 4, ...: the four waves of a block pull adjacent half lines, so every 128-B line is
//    used whole within one block);
//  * operands go straight from global memory into MFMA fragments (global_load_dwordx4, 16 B per lane = the 16x16x32
//    fragment of a lane: row lane & 15, k-chunk lane >> 4) — no LDS ring, no barrier in the K loop, two chunks of G steps
//    double-buffered in registers (2 x G x (2 MT + 2 NT) loads in flight per wave);
//  * the four partial accumulators meet in LDS and are summed in wave order (deterministic), then the usual epilogue.
// Arithmetic: the same three products per step as gemm_kernel.h's SPLIT = 1 (hi*hi + lo_w*hi_a + hi_w*lo_a; SPW = 1 drops
// the zero lo_w product, the f16x2 form); the K order differs (four interleaved partial sums), i.e. fp32 re-association.
#pragma once
#include "gemm_kernel.h"

template <int MT, int NT, int SPW>
__global__ __launch_bounds__(256, 2) void gemm_skinny_x3_kernel(GemmArgs p, int out_kind) {
  constexpr int NW = 4, G = 3;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fk = lane >> 4;
  const int n0 = blockIdx.x * (NT * 16), m0 = blockIdx.y * (MT * 16), batch = blockIdx.z;
  const half_t* A = p.A + (long)batch * p.sA;
  const half_t* W = p.W + (long)batch * p.sW;
  const half_t* ap[MT];
  const half_t* wp[NT];
#pragma unroll
  for (int t = 0; t < MT; ++t) ap[t] = A + (long)min(m0 + t * 16 + frow, p.M - 1) * p.lda + fk * 8;
#pragma unroll
  for (int t = 0; t < NT; ++t) wp[t] = W + (long)min(n0 + t * 16 + frow, p.N - 1) * p.ldw + fk * 8;

  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // the epilogue's operands of the sub-tile this wave will finish (wave w finishes sub-tile w) are requested NOW, in front of the K loop:
  // behind it they are two more dependent memory latencies of a launch that is little else (~1 us of a 9-us launch)
  static_assert(MT * NT == NW, "one sub-tile per wave");
  const long cb = (long)batch * p.sC;
  const float* R = p.R ? p.R + (long)batch * p.sR : nullptr;
  const int own_nt = wave / MT, own_mt = wave - own_nt * MT;
  const int own_m = m0 + own_mt * 16 + frow, own_n = n0 + own_nt * 16 + fk * 4;
  const bool own_ok = own_m < p.M && own_n < p.N;
  f32x4 bias_v = {0.f, 0.f, 0.f, 0.f}, res_v = {0.f, 0.f, 0.f, 0.f};
  if (p.vec_ok && own_ok) {
    if (p.bias) bias_v = *(const f32x4*)(p.bias + own_n);
    if (R) res_v = *(const f32x4*)(R + (long)(own_m % p.res_rows) * p.ldr + own_n);
  }

  const int nk = p.K / BK;                                   // 32-wide steps; this wave: wave, wave + NW, ...
  const int mine = nk > wave ? (nk - wave + NW - 1) / NW : 0;
  const int nchunks = (mine + G - 1) / G;
  struct Frag { half8_t ah[G][MT], al[G][MT], wh[G][NT], wl[G][SPW == 2 ? NT : 1]; };
  auto load = [&](Frag& f, int c) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      int s = c * G + g;
      s = s < mine ? s : mine - 1;                           // branch-free: a surplus step re-reads the last one (its MFMAs are skipped)
      const long ko = (long)(wave + s * NW) * BK;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        f.ah[g][t] = *(const half8_t*)(ap[t] + ko);
        f.al[g][t] = *(const half8_t*)(ap[t] + p.planeA + ko);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        f.wh[g][t] = *(const half8_t*)(wp[t] + ko);
        if constexpr (SPW == 2) f.wl[g][t] = *(const half8_t*)(wp[t] + p.planeW + ko);
      }
    }
  };
  auto compute = [&](const Frag& f, int c) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (c * G + g < mine) {                                // wave-uniform
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.wh[g][nt], f.ah[g][mt], acc[nt][mt], 0, 0, 0);
        if constexpr (SPW == 2) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.wl[g][nt], f.ah[g][mt], acc[nt][mt], 0, 0, 0);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.wh[g][nt], f.al[g][mt], acc[nt][mt], 0, 0, 0);
      }
    }
  };
  if (mine > 0) {
    Frag f0, f1;
    load(f0, 0);
    for (int c = 0; c < nchunks; c += 2) {
      if (c + 1 < nchunks) load(f1, c + 1);
      compute(f0, c);
      if (c + 2 < nchunks) load(f0, c + 2);
      if (c + 1 < nchunks) compute(f1, c + 1);
    }
  }

  // ---- the four partial tiles meet in LDS; wave w finishes sub-tiles w, w + NW, ... (sum in wave order: deterministic)
  constexpr int NTILE = MT * NT;
  __shared__ f32x4 red[NW][NTILE][64];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) red[wave][nt * MT + mt][lane] = acc[nt][mt];
  __syncthreads();
  {
    const int t = wave;
    f32x4 v = red[0][t][lane];
#pragma unroll
    for (int w = 1; w < NW; ++w) v += red[w][t][lane];
    const int m = own_m, n = own_n;
    if (!own_ok) return;
    v *= p.out_scale;
    const long ci = cb + (long)m * p.ldc + n;
    if (p.vec_ok) {                                          // N % 4 == 0 and 16-byte aligned rows: a lane's four columns are in or out together
      if (p.bias) v += bias_v;
      if (p.act != ZH_ACT_NONE) { v[0] = zh_act(v[0], p.act); v[1] = zh_act(v[1], p.act); v[2] = zh_act(v[2], p.act); v[3] = zh_act(v[3], p.act); }
      if (R) v += res_v;
      if (out_kind == 0) *(f32x4*)((float*)p.C + ci) = v;
      else zh_store_h4((half_t*)p.C + ci, out_kind == 2 ? p.planeC : 0, v);
    } else {                                                 // ragged N / unaligned rows: element by element
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (n + e >= p.N) break;
        float x = v[e];
        if (p.bias) x += p.bias[n + e];
        x = zh_act(x, p.act);
        if (R) x += R[(long)(m % p.res_rows) * p.ldr + n + e];
        if (out_kind == 0) ((float*)p.C)[ci + e] = x;
        else zh_store_h1((half_t*)p.C + ci + e, out_kind == 2 ? p.planeC : 0, x);
      }
    }
  }
}

// Host side: is this GEMM one for the few-row kernel?  Shape only (never alignment): the K order of a row's sum — and with it the bits of
// the result — must not depend on how many columns a caller asks for (sharded retrieval merges shards of any width).
static inline bool gemm_skinny_ok(const GemmArgs& p, int batch, bool vec_ok, int max_rows) {
  (void)vec_ok;                                              // the epilogue has an element-wise form: every few-row GEMM takes this kernel
  return !p.pos_y && p.M <= max_rows && (long)zh_cdiv(p.N, 32) * zh_cdiv(p.M, 32) * batch <= 65535L * 8;
}

template <int SPW>
static void launch_skinny(const GemmArgs& p, int batch, int out_kind, hipStream_t stream) {
  const dim3 grid(zh_cdiv(p.N, 32), zh_cdiv(p.M, 32), batch);
  hipLaunchKernelGGL((gemm_skinny_x3_kernel<2, 2, SPW>), grid, dim3(256), 0, stream, p, out_kind);
}
