// fp16-input / fp32-accumulate MFMA GEMM with fused epilogues (gfx950).
//
//   C[b][m][n] = act( sum_k A[b][m][k] * W[b][n][k] + bias[n] ) + R[b][m % res_rows][n]
//
// Both operands are K-contiguous ("NT" form = torch Linear layout, networks/clip_arch.py:304-310),
// so one kernel family serves every contraction on the hot path: patch-embed conv-as-GEMM
// (clip_arch.py:378), QKV/out-proj/MLP (clip_arch.py:314-320), ffn1/ffn2 (zutis.py:546-549),
// decoder projections/FFN (transformer.py:272-290), the mask einsum (zutis.py:196-198, batched, sigmoid
// epilogue), the text-space projection (zutis.py:319) and the class-logit einsum (zutis.py:361-365).
//
// Design (MI355X):
//  * v_mfma_f32_16x16x32_f16; operand roles swapped (MFMA-A = W rows, MFMA-B = A rows) so a lane's 4
//    accumulator registers are 4 consecutive n of one output row -> 16-byte row-major stores and float4
//    bias/residual loads.
//  * Block tiles 256x256 / 256x192 (8 waves, 2x4) for the big GEMMs, 128x128 (4 waves) for small ones; the
//    first profile showed the 128x128x64 double-buffered version was load-LATENCY bound (K-step time == loaded
//    L2 latency ~1.2 us), so the K loop is now a 4-stage ring of BK=32 slices: HBM/L2 -> LDS by
//    global_load_lds_dwordx4 issued THREE slices ahead, retired by a counted s_waitcnt vmcnt(N) + one raw
//    s_barrier per slice (never vmcnt(0) in the steady state).  Per-lane source pointers are advanced by a
//    constant, so a slice costs 4 DMA issues + 4 pointer adds per wave.
//  * LDS image is lane-linear (64-byte rows, 16 rows per 1-KiB DMA piece); ds_read_b128 bank conflicts are
//    removed by XOR-ing the 16-byte chunk index with (-(row>>2))&3 on the DMA *source* address and on the read
//    address (conflict-free for the 16x16x32 operand lane groups).
//  * Block ids: XCD-aware bijective remap, then 8-row super-tiles so one XCD's concurrent tiles share panels in
//    its 4 MiB L2 (measured L2 hit rate 82 %).
#include "common.h"

#define BK 32
#define GROUP_M 8

struct GemmArgs {
  const half_t* A; long lda, sA;
  const half_t* W; long ldw, sW;
  void* C; long ldc, sC;
  const float* bias;
  const float* R; long ldr, sR; int res_rows;
  int M, N, K, act, nbm, nbn, vec_ok, group_m;
#ifdef ZH_GEMM_PROBE
  long long* probe;   // developer build (tools/gemm_probe.py): 4 timestamps per block
#endif
};
#ifdef ZH_GEMM_PROBE
static long long* g_probe = nullptr;
extern "C" void zh_gemm_set_probe(long long* p) { g_probe = p; }
#define ZH_PROBE(i) do { if (p.probe && tid == 0) { p.probe[(long)blockIdx.x * 8 + (i)] = wall_clock64(); p.probe[(long)blockIdx.x * 8 + 4 + (i)] = clock64(); } } while (0)
#else
#define ZH_PROBE(i)
#endif

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt_barrier() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}
// runtime count of whole stages (0 .. MAXC) that may stay in flight, NP DMA issues each
template <int NP, int MAXC>
__device__ __forceinline__ void wait_stages_barrier(int c) {
  static_assert(MAXC <= 5, "extend the switch");
  switch (c) {
    case 0: wait_vmcnt_barrier<0>(); break;
    case 1: wait_vmcnt_barrier<NP>(); break;
    case 2: if (MAXC >= 2) { wait_vmcnt_barrier<(MAXC >= 2 ? 2 : 0) * NP>(); break; }
    case 3: if (MAXC >= 3) { wait_vmcnt_barrier<(MAXC >= 3 ? 3 : 0) * NP>(); break; }
    case 4: if (MAXC >= 4) { wait_vmcnt_barrier<(MAXC >= 4 ? 4 : 0) * NP>(); break; }
    default: wait_vmcnt_barrier<MAXC * NP>(); break;
  }
}

// WM x WN waves; each wave owns TM x TN subtiles of 16x16.  Block tile = (WM*TM*16) x (WN*TN*16).
// STAGES = depth of the LDS ring: 4 for the big tiles; 8 for the small-tile variants used when a GEMM has fewer tiles than
// the chip has CUs — those are bound by bytes in flight per CU (3 x 16 KiB per 128x128 block = 24 GB/s per CU at ~2 us of
// loaded latency), so the ring, not the tile, is what has to grow.
template <int WM, int WN, int TM, int TN, int STAGES, int OUT_F16, int ACT, int VEC>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN) >= 8 ? 2 : ((WM * WN * TM * TN) >= 96 ? 2 : 1)) void gemm_f16_kernel(GemmArgs p) {
  constexpr int NW = WM * WN;
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  constexpr int ROWS = BM + BN;             // LDS rows per stage (A rows then W rows), 64 B each
  constexpr int PIECES = ROWS / 16;         // 1-KiB DMA pieces per stage
  constexpr int NP = (PIECES + NW - 1) / NW;  // DMA issues per wave per stage (duplicates pad uneven splits)
  static_assert(NP >= 2 && NP <= 8 && STAGES >= 4 && STAGES <= 8, "unsupported pieces-per-wave count / ring depth");
  constexpr int AHEAD = STAGES - 3;           // whole stages that may still be in flight at a steady-state barrier
  constexpr int STAGE_HALVES = ROWS * BK;
  __shared__ __attribute__((aligned(16))) half_t smem[STAGES * STAGE_HALVES];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WN, wc = wave % WN;
  ZH_PROBE(0);

  // XCD-aware bijective remap: blocks b, b+8, ... share an XCD -> give each XCD a contiguous id range
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tiles = p.nbm * p.nbn;
  const int batch = wg / tiles;
  const int trem = wg - batch * tiles;
  // super-tile order: GROUP_M consecutive ids walk GROUP_M m-tiles of one n-tile
  const int gsz = p.group_m * p.nbn;
  const int gid = trem / gsz;
  const int gfirst = gid * p.group_m;
  const int grows = min(p.nbm - gfirst, p.group_m);
  const int gl = trem - gid * gsz;
  const int tm = gfirst + gl % grows, tn = gl / grows;
  const int m0 = tm * BM, n0 = tn * BN;

  const half_t* A = p.A + (long)batch * p.sA;
  const half_t* W = p.W + (long)batch * p.sW;

  // per-lane DMA source pointers: piece pc covers LDS rows [16*pc, 16*pc+16); lane -> row (lane>>2), phys chunk lane&3
  const half_t* gp[NP];
  int lds_piece[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    int pc = wave + i * NW;
    pc = pc < PIECES ? pc : PIECES - 1;
    lds_piece[i] = pc * 16 * BK;
    const int R = pc * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((-(R >> 2)) & 3);
    if (R < BM) {
      int row = m0 + R;
      row = row < p.M ? row : p.M - 1;
      gp[i] = A + (long)row * p.lda + c * 8;
    } else {
      int row = n0 + (R - BM);
      row = row < p.N ? row : p.N - 1;
      gp[i] = W + (long)row * p.ldw + c * 8;
    }
  }
  auto issue_stage = [&](int slot) {
    half_t* sb = smem + slot * STAGE_HALVES;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      __builtin_amdgcn_global_load_lds((glb_ptr_t)gp[i], (lds_ptr_t)(sb + lds_piece[i]), 16, 0, 0);
      gp[i] += BK;
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;                    // even: K % 64 == 0
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nk) issue_stage(s);

  const int frow = lane & 15, fk = lane >> 4;
  const int foff = frow * BK + ((fk ^ ((-(frow >> 2)) & 3)) * 8);   // per-lane offset inside a 16-row subtile
  const half_t* rdA = smem + (wr * TM * 16) * BK + foff;
  const half_t* rdW = smem + (BM + wc * TN * 16) * BK + foff;

  // Register double-buffered fragments: while the MFMAs of slice kt run, the ds_read_b128 of slice kt+1 are in
  // flight (the LDS latency at the head of every slice was exposed on all 8 waves at once behind the barrier).
  // Slice kt+1 must therefore have landed one iteration earlier: counted waits are vmcnt(4) in the steady state.
  half8_t fa0[TM], fw0[TN], fa1[TM], fw1[TN];
  auto load_frags = [&](int kt, half8_t (&fa)[TM], half8_t (&fw)[TN]) {
    const int so = (kt % STAGES) * STAGE_HALVES;
#pragma unroll
    for (int t = 0; t < TM; ++t) fa[t] = *(const half8_t*)(rdA + so + t * 16 * BK);
#pragma unroll
    for (int t = 0; t < TN; ++t) fw[t] = *(const half8_t*)(rdW + so + t * 16 * BK);
  };
  auto mfma_all = [&](half8_t (&fa)[TM], half8_t (&fw)[TN]) {
#pragma unroll
    for (int nt = 0; nt < TN; ++nt)
#pragma unroll
      for (int mt = 0; mt < TM; ++mt)
        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[nt], fa[mt], acc[nt][mt], 0, 0, 0);
  };
  // steady-state phase (kt + 3 < nk): branch-free so the scheduler can interleave — every group of MFMAs shadows one
  // LDS fragment read or one LDS-DMA issue of the NEXT slices; the wave's stream stays MFMA-paced instead of
  // front-loading 16 memory instructions behind the barrier.
  auto steady = [&](int kt, half8_t (&fa)[TM], half8_t (&fw)[TN], half8_t (&na)[TM], half8_t (&nw)[TN]) {
#ifndef ZH_X_NOBAR
    wait_vmcnt_barrier<AHEAD * NP>();
#endif
    // program order = dependence order for the compiler: an LDS-DMA is a write to `smem`, so fragment reads placed after
    // it can never be scheduled above it.  Reads first, DMA second lets the reads spread under the first MFMAs and the DMA
    // issues under the last ones (the other order left all 12 ds_read_b128 + their latency exposed at the end of the slice).
#ifndef ZH_X_NOFRAG
    load_frags(kt + 1, na, nw);
#endif
#ifndef ZH_X_NODMA
    issue_stage((kt + STAGES - 1) % STAGES);
#endif
    mfma_all(fa, fw);
    constexpr int NMEM = TM + TN + NP, NMFMA = TM * TN;
    if (NMFMA >= NMEM) {
#pragma unroll
      for (int i = 0; i < TM + TN; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, NMFMA / NMEM, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, NMFMA / NMEM, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    }
  };
  auto tail = [&](int kt, half8_t (&fa)[TM], half8_t (&fw)[TN], half8_t (&na)[TM], half8_t (&nw)[TN]) {
    if (kt + 1 < nk) {
      // slices issued so far: 0 .. min(nk-1, kt+STAGES-2); slice kt+1 must have landed, the later ones may fly
      wait_stages_barrier<NP, AHEAD>(min(nk - 1, kt + STAGES - 2) - (kt + 1));
      load_frags(kt + 1, na, nw);                                          // reads before the DMA: see steady()
      if (kt + STAGES - 1 < nk) issue_stage((kt + STAGES - 1) % STAGES);
    }
    mfma_all(fa, fw);   // (a sched_group_barrier interleave here makes hipcc spill: 172 scratch ops, 2.6x slower)
  };
  if (nk >= STAGES - 1) wait_vmcnt_barrier<(STAGES - 2) * NP>();   // stage 0 landed; the other STAGES-2 may still be in flight
  else wait_vmcnt_barrier<0>();                                     // short K: not worth a counted wait
  ZH_PROBE(1);
  load_frags(0, fa0, fw0);
  int kt = 0;
  for (; kt + STAGES < nk; kt += 2) {
    steady(kt, fa0, fw0, fa1, fw1);
    steady(kt + 1, fa1, fw1, fa0, fw0);
  }
  for (; kt < nk; kt += 2) {
    tail(kt, fa0, fw0, fa1, fw1);
    tail(kt + 1, fa1, fw1, fa0, fw0);
  }

  ZH_PROBE(2);
  // ---- epilogue: lane owns rows m = ..+(lane&15), 4 consecutive n at 4*(lane>>4).  ACT / VEC are template
  // parameters: a runtime switch unrolled 32x blew the instruction cache (fc GEMM 1.4x slower in the model).
  const long cb = (long)batch * p.sC;
  const float* R = p.R ? p.R + (long)batch * p.sR : nullptr;
  if (VEC == 2) {
    // LDS-staged epilogue: the direct form stores 32-byte runs (4 lanes x 8 B) into 16 different 128-B lines per
    // instruction and measured 2.4 TB/s, fully exposed (34 % of a K=768 tile).  Here each wave transposes its tile through
    // a private, conflict-free LDS slab (row stride +16 B) and writes whole rows with 16 B per lane.
    constexpr int ESZ = OUT_F16 ? 2 : 4;
    constexpr int RS = TN * 16 * ESZ + 16;                  // slab row stride (bytes)
    constexpr int PR = (OUT_F16 ? 64 : 32) < TM * 16 ? (OUT_F16 ? 64 : 32) : TM * 16;   // rows per pass
    constexpr int MTP = PR / 16;
    constexpr int CPRW = TN * 16 * ESZ / 16;                // 16-B chunks per row
    constexpr int NIT = PR * CPRW / 64;
    static_assert((PR * CPRW) % 64 == 0, "epilogue slab must divide into full wave reads");
    static_assert(NW * PR * RS <= (int)sizeof(smem), "epilogue slabs exceed the ring");
    __syncthreads();                                        // ring no longer read; every LDS-DMA has landed
    char* slab = (char*)smem + wave * (PR * RS);
#pragma clang loop unroll(full)
    for (int pass = 0; pass < TM / MTP; ++pass) {
#pragma clang loop unroll(full)
      for (int nt = 0; nt < TN; ++nt) {
        const int n = n0 + (wc * TN + nt) * 16 + fk * 4;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && n < p.N) bv = *(const f32x4*)(p.bias + n);
#pragma clang loop unroll(full)
        for (int ml = 0; ml < MTP; ++ml) {
          f32x4 v = acc[nt][pass * MTP + ml] + bv;
          if (ACT != ZH_ACT_NONE) {
            v[0] = zh_act(v[0], ACT); v[1] = zh_act(v[1], ACT); v[2] = zh_act(v[2], ACT); v[3] = zh_act(v[3], ACT);
          }
          char* dst = slab + (ml * 16 + frow) * RS + (nt * 16 + fk * 4) * ESZ;
          if (OUT_F16) {
            half4_t h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            *(half4_t*)dst = h;
          } else {
            *(f32x4*)dst = v;
          }
        }
      }
#pragma clang loop unroll(full)
      for (int it = 0; it < NIT; ++it) {
        const int c = it * 64 + lane;
        const int row = c / CPRW, ch = c - row * CPRW;
        const int m = m0 + wr * TM * 16 + pass * PR + row;
        const int n = n0 + wc * TN * 16 + ch * (16 / ESZ);
        f32x4 d = *(const f32x4*)(slab + row * RS + ch * 16);
        if (m < p.M && n < p.N) {
          if (!OUT_F16 && R) d += *(const f32x4*)(R + (long)(m % p.res_rows) * p.ldr + n);
if (OUT_F16) *(f32x4*)((half_t*)p.C + cb + (long)m * p.ldc + n) = d;
          else *(f32x4*)((float*)p.C + cb + (long)m * p.ldc + n) = d;
        }
      }
    }
  } else if (VEC) {
#pragma clang loop unroll(full)
    for (int nt = 0; nt < TN; ++nt) {
      const int n = n0 + (wc * TN + nt) * 16 + fk * 4;
      const bool nok = n < p.N;
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (p.bias && nok) bv = *(const f32x4*)(p.bias + n);
#pragma clang loop unroll(full)
      for (int mt = 0; mt < TM; ++mt) {
        const int m = m0 + (wr * TM + mt) * 16 + frow;
        if (nok && m < p.M) {
          f32x4 v = acc[nt][mt] + bv;
          if (ACT != ZH_ACT_NONE) {
            v[0] = zh_act(v[0], ACT); v[1] = zh_act(v[1], ACT); v[2] = zh_act(v[2], ACT); v[3] = zh_act(v[3], ACT);
          }
          if (R) v += *(const f32x4*)(R + (long)(m % p.res_rows) * p.ldr + n);
          if (OUT_F16) {
            half4_t h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            *(half4_t*)((half_t*)p.C + cb + (long)m * p.ldc + n) = h;
          } else {
            *(f32x4*)((float*)p.C + cb + (long)m * p.ldc + n) = v;
          }
        }
      }
    }
  } else {   // unaligned / odd-N fallback: scalar stores (rare: odd pixel counts)
#pragma clang loop unroll(full)
    for (int mt = 0; mt < TM; ++mt) {
      const int m = m0 + (wr * TM + mt) * 16 + frow;
      const long rrow = R ? (long)(m % p.res_rows) * p.ldr : 0;
#pragma clang loop unroll(full)
      for (int nt = 0; nt < TN; ++nt) {
        const int n = n0 + (wc * TN + nt) * 16 + fk * 4;
#pragma clang loop unroll(full)
        for (int e = 0; e < 4; ++e) {
          if (m < p.M && n + e < p.N) {
            float x = acc[nt][mt][e];
            if (p.bias) x += p.bias[n + e];
            x = zh_act(x, ACT);
            if (R) x += R[rrow + n + e];
            if (OUT_F16) ((half_t*)p.C)[cb + (long)m * p.ldc + n + e] = (half_t)x;
            else ((float*)p.C)[cb + (long)m * p.ldc + n + e] = x;
          }
        }
      }
    }
  }
#ifdef ZH_GEMM_PROBE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ZH_PROBE(3);
#endif
}


template <int WM, int WN, int TM, int TN, int STAGES, int OUT_F16, int ACT, int VEC>
static void launch_one(GemmArgs p, int batch, hipStream_t stream) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  p.nbm = zh_cdiv(p.M, BM);
  p.nbn = zh_cdiv(p.N, BN);
  const unsigned nblk = (unsigned)((long)p.nbm * p.nbn * batch);
  hipLaunchKernelGGL((gemm_f16_kernel<WM, WN, TM, TN, STAGES, OUT_F16, ACT, VEC>), dim3(nblk), dim3(64 * WM * WN), 0, stream, p);
}

// Instantiated (out type, activation) pairs = the ones the hot path uses; anything else is an argument error.
template <int WM, int WN, int TM, int TN, int STAGES, int VEC>
static bool launch_gemm(const GemmArgs& p, int batch, int out_f16, hipStream_t stream) {
  const int key = out_f16 * 8 + p.act;
  switch (key) {
    case 0 + ZH_ACT_NONE: launch_one<WM, WN, TM, TN, STAGES, 0, ZH_ACT_NONE, VEC>(p, batch, stream); return true;
    case 0 + ZH_ACT_SIGMOID: launch_one<WM, WN, TM, TN, STAGES, 0, ZH_ACT_SIGMOID, VEC>(p, batch, stream); return true;
    case 8 + ZH_ACT_NONE: launch_one<WM, WN, TM, TN, STAGES, 1, ZH_ACT_NONE, VEC>(p, batch, stream); return true;
    case 8 + ZH_ACT_QUICKGELU: launch_one<WM, WN, TM, TN, STAGES, 1, ZH_ACT_QUICKGELU, VEC>(p, batch, stream); return true;
    case 8 + ZH_ACT_RELU: launch_one<WM, WN, TM, TN, STAGES, 1, ZH_ACT_RELU, VEC>(p, batch, stream); return true;
    case 8 + ZH_ACT_GELU_ERF: launch_one<WM, WN, TM, TN, STAGES, 1, ZH_ACT_GELU_ERF, VEC>(p, batch, stream); return true;
    default: return false;
  }
}

// Relative time estimate of a tiling: rounds of the 256-CU chip x time of one round.  With `bpc` blocks resident
// per CU a round takes bpc x the tile's own time; `eff` is the measured relative speed of the tile shape at
// K=768 (256x256: 1.0, 256x192: 0.95, 128x128: 0.8 — tools/gemm_bench.py on MI355X).
static double tiling_cost(long M, long N, int batch, int BM, int BN, int bpc, double eff) {
  const long tiles = (long)zh_cdiv(M, BM) * zh_cdiv(N, BN) * batch;
  const long slots = 256L * bpc;
  const long rounds = (tiles + slots - 1) / slots;
  return (double)rounds * BM * BN * bpc / eff;
}

extern "C" int zh_gemm_f16(const void* A, long lda, long strideA, const void* W, long ldw, long strideW,
                           void* C, long ldc, long strideC, int out_f16,
                           const float* bias, const float* residual, long ldr, long strideR, int res_rows,
                           int act, int M, int N, int K, int batch, hipStream_t stream) {
  ZH_CHECK_ARG(A && W && C, "zh_gemm_f16: null operand");
  ZH_CHECK_ARG(M > 0 && N > 0 && K > 0 && batch > 0, "zh_gemm_f16: bad shape M=%d N=%d K=%d batch=%d", M, N, K, batch);
  ZH_CHECK_ARG(K % 64 == 0, "zh_gemm_f16: K=%d must be a multiple of 64", K);
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
  p.M = M; p.N = N; p.K = K; p.act = act; p.nbm = p.nbn = 0;
  // super-tile height (developer override ZH_GEMM_GROUP_M): 3..8 measure within 1 % of each other on the model, 16 / 32 lose
  // 13 / 36 % on 8192^3 (tools/gemm_vs_blaslt.py) — the order in which an XCD's 32 resident tiles share panels matters
  { const char* g = getenv("ZH_GEMM_GROUP_M"); p.group_m = g ? atoi(g) : GROUP_M; if (p.group_m < 1) p.group_m = GROUP_M; }
#ifdef ZH_GEMM_PROBE
  p.probe = g_probe;
#endif
  const int esz = out_f16 ? 2 : 4;
  p.vec_ok = (N % 4 == 0) && (ldc % 4 == 0) && (strideC % 4 == 0) && (((uintptr_t)C & (4 * esz - 1)) == 0) &&
             (!bias || ((uintptr_t)bias & 15) == 0) &&
             (!residual || (ldr % 4 == 0 && strideR % 4 == 0 && ((uintptr_t)residual & 15) == 0));
  ZH_CHECK_ARG((long)zh_cdiv(M, 128) * zh_cdiv(N, 128) * batch < (1L << 31), "zh_gemm_f16: grid too large");
  const char* force = getenv("ZH_GEMM_TILE");   // developer override: 128 | 192 | 256
  const double c256 = tiling_cost(M, N, batch, 256, 256, 1, 1.0);
  const double c192 = tiling_cost(M, N, batch, 256, 192, 1, 0.95);
  const double c128 = tiling_cost(M, N, batch, 128, 128, 2, 0.8);
  int pick = (c128 < c256 && c128 < c192) ? 128 : (c192 < c256 ? 192 : 256);
  // (a 128x64 tile for the decoder's M = B*Q GEMMs measured no better than 128x128: they are slice-latency bound;
  //  it stays reachable through ZH_GEMM_TILE=64 for experiments)
  // few-tile GEMMs (batch-1 inference: M = 442 tokens -> 24 tiles of 128x128 on 256 CUs): 64x64 tiles on an 8-deep
  // ring put 4x the CUs to work; measured 32 -> 16 us on the 442x768x3072 MLP projection (tools/gemm_small.py)
  if ((long)zh_cdiv(M, 128) * zh_cdiv(N, 128) * batch <= 96) pick = 3064;
  if (force) pick = atoi(force);
  // 16-byte row stores need 16-B aligned rows; an f16 residual is not supported (none on the hot path)
  const bool wide_ok = p.vec_ok && (((uintptr_t)C & 15) == 0) && ((ldc * esz) % 16 == 0) && ((strideC * esz) % 16 == 0) &&
                       ((N * esz) % 16 == 0) && !(out_f16 && residual);
  bool ok;
  if (!p.vec_ok) ok = launch_gemm<2, 2, 4, 4, 4, 0>(p, batch, out_f16, stream);      // scalar-store fallback: small tile only
  else if (!wide_ok) ok = launch_gemm<2, 2, 4, 4, 4, 1>(p, batch, out_f16, stream);  // direct 8/16-B stores
  else if (pick == 2128) ok = launch_gemm<2, 2, 4, 4, 8, 2>(p, batch, out_f16, stream);   // 128 x 128, 8-deep ring
  else if (pick == 2064) ok = launch_gemm<2, 2, 4, 2, 8, 2>(p, batch, out_f16, stream);   // 128 x 64, 8-deep ring
  else if (pick == 3064) ok = launch_gemm<2, 2, 2, 2, 8, 2>(p, batch, out_f16, stream);   // 64 x 64, 8-deep ring
  else if (pick == 64) ok = launch_gemm<2, 2, 4, 2, 4, 2>(p, batch, out_f16, stream);   // 128 x 64
  else if (pick == 128) ok = launch_gemm<2, 2, 4, 4, 4, 2>(p, batch, out_f16, stream);
  else if (pick == 192) ok = launch_gemm<2, 4, 8, 3, 4, 2>(p, batch, out_f16, stream);
  else if (pick == 1192) ok = launch_gemm<2, 2, 4, 6, 4, 2>(p, batch, out_f16, stream);   // 128 x 192, 4 waves, 2 blocks/CU: experiment only (ZH_GEMM_TILE), measured 10-20 % slower than 256 x 192 on the N=768 GEMMs
  else ok = launch_gemm<2, 4, 8, 4, 4, 2>(p, batch, out_f16, stream);
  ZH_CHECK_ARG(ok, "zh_gemm_f16: (out_f16=%d, act=%d) is not an instantiated epilogue (f32: none|sigmoid; f16: none|quickgelu|relu|gelu_erf)",
               out_f16, act);
  ZH_CHECK_LAUNCH("zh_gemm_f16");
  return ZH_OK;
}
