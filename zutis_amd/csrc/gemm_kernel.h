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
//  * SPLIT = 1 ("f16x3", the reference-equivalent mode): both operands arrive as a pair of fp16 planes hi = f16(x),
//    lo = f16(x - hi) (22 significand bits), a K slice stages four row sets (A hi, A lo, W hi, W lo) and every
//    accumulator gets three MFMAs, hi*hi + lo_w*hi_a + hi_w*lo_a, in fp32 (the dropped lo*lo term is 2^-22 relative).
//    1.5x the MFMA work per staged byte of the plain kernel; 3-slot ring with prefetch distance 3.
//  * SPLIT = 2 ("f16x2"): the same mode for a weight whose lo plane is ZERO — every value of W * 2^s is an fp16 number, which
//    is what the reference's own constructor path produces for the released CLIP towers (`convert_weights` rounds every
//    Linear / conv / attention weight to fp16 before ZUTIS casts the encoder back to fp32, clip_arch.py:566-587,625 and
//    zutis.py:55; the index-dataset pipeline, utils/extract_image_embeddings.py, only ever runs such weights).  The W lo rows are
//    neither staged nor multiplied: two MFMAs per accumulator, hi_w*hi_a + hi_w*lo_a, in the order the SPLIT = 1 kernel issues
//    its non-zero products — the results are bit-identical to SPLIT = 1 on a zero lo plane (tests/test_kernels_gpu.py), with
//    2/3 of the MFMA work and 3/4 of the staged bytes.  The host picks it per weight at pack time (ops.split_weight).
//  * Block ids: XCD-aware bijective remap, then 4-row super-tiles so one XCD's concurrent tiles share panels in
//    its 4 MiB L2 (measured L2 hit rate 82 %; 4 rows: 5 % less fabric traffic than 8 and +1 % / +3 % on the fast / exact step).
#pragma once
#include "common.h"
#include <stdlib.h>
#include <type_traits>

#define BK 32
#define GROUP_M 4

struct GemmArgs {
  const half_t* A; long lda, sA;
  const half_t* W; long ldw, sW;
  void* C; long ldc, sC;
  long planeA, planeW, planeC;   // SPLIT: element offset hi plane -> lo plane of A / W / (OUT == 2) C
  float out_scale;               // SPLIT: accumulators are multiplied by this before the bias (weights packed as W * 2^s)
  const float* bias;
  const float* R; long ldr, sR; int res_rows;
  // separable per-pixel row bias (rows are pixels m = img * pos_hw + y * pos_w + x): pos_y[y][n] + pos_x[x][n] is added to
  // the accumulator before the activation — the tile's accumulators START from it, loaded under the ring's prologue
  const void* pos_y; const void* pos_x; long ld_pos; int pos_hw, pos_w, pos_f16;   // tables fp32, or fp16 (pos_f16)
  int M, N, K, act, nbm, nbn, vec_ok, group_m;
  int total;                     // tiles x batch items of the launch (the persistent big tiles walk them with stride gridDim.x)
#ifdef ZH_GEMM_PROBE
  long long* probe;   // developer build (tools/gemm_probe.py): 4 timestamps per block
#endif
};
#ifdef ZH_GEMM_PROBE
static long long* g_probe = nullptr;
#ifdef ZH_GEMM_MAIN
extern "C" void zh_gemm_set_probe(long long* p) { g_probe = p; }
#endif
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
  static_assert(MAXC <= 9, "extend the switch");
  switch (c) {
    case 0: wait_vmcnt_barrier<0>(); break;
    case 1: wait_vmcnt_barrier<NP>(); break;
    case 2: if (MAXC >= 2) { wait_vmcnt_barrier<(MAXC >= 2 ? 2 : 0) * NP>(); break; }
    case 3: if (MAXC >= 3) { wait_vmcnt_barrier<(MAXC >= 3 ? 3 : 0) * NP>(); break; }
    case 4: if (MAXC >= 4) { wait_vmcnt_barrier<(MAXC >= 4 ? 4 : 0) * NP>(); break; }
    case 5: if (MAXC >= 5) { wait_vmcnt_barrier<(MAXC >= 5 ? 5 : 0) * NP>(); break; }
    case 6: if (MAXC >= 6) { wait_vmcnt_barrier<(MAXC >= 6 ? 6 : 0) * NP>(); break; }
    case 7: if (MAXC >= 7) { wait_vmcnt_barrier<(MAXC >= 7 ? 7 : 0) * NP>(); break; }
    case 8: if (MAXC >= 8) { wait_vmcnt_barrier<(MAXC >= 8 ? 8 : 0) * NP>(); break; }
    default: wait_vmcnt_barrier<MAXC * NP>(); break;
  }
}

// Split-pair tiles whose 64-k slice is more than a third of the LDS: a CIRCULAR ring of 1-KiB pieces instead of whole slots (the value
// is its size in pieces; 0 = whole slots).  A 128 x 96 tile stages 56 KiB per slice — two slots, ONE slice in flight; as a 160-piece
// circle the stream runs 48 .. 104 KiB ahead of the slice being read: the space of slice kt - 1 is refilled, behind barrier kt, with the
// last pieces of slice kt + 1 and the first of slice kt + 2.  (`stages` only tells these instantiations from the whole-slot ones.)
constexpr int gemm_k64_ring_pieces(int wm, int wn, int tm, int tn, int stages, int split) {
  return split != 1 ? 0                                                                   // (the two-plane-W form only: x2 slices are a quarter smaller and fit three whole slots)
       : (wm == 4 && wn == 2 && tm == 2 && tn == 3 && stages == 4) ? 160                  // 128 x 96, 8 waves of 32 x 48: 2.86 slices
       : (wm == 4 && wn == 2 && tm == 2 && tn == 4 && stages == 3) ? 160                  // 128 x 128, 8 waves of 32 x 64: 2.5 slices
       : 0;
}
// Split-pair tiles that take 64-k slices on MORE than two slots (every other (tile, ring depth) pair below the big tiles is on 32-k
// slices; two slots always mean 64-k).  Round 5: with ONE slice of prefetch every slice pays the whole load latency (the K loop of the
// two-slot form ran 1.5 us per slice against 0.66 us of stream time); these keep two or more slices in flight —
// whole slots, or the circular ring above.
constexpr bool gemm_k64_deep(int wm, int wn, int tm, int tn, int stages, int split) {
  return (wm == 4 && wn == 2 && tm == 2 && tn == 2 && stages == 3)                        // 128 x 64, 8 waves of 32 x 32, 3 x 48 KiB
      || (wm == 2 && wn == 2 && tm == 2 && tn == 2 && (stages == 4 || stages == 5))       // 64 x 64, 4 waves, 4 - 5 x 32 KiB
      || gemm_k64_ring_pieces(wm, wn, tm, tn, stages, split) != 0;
}

// Plain-fp16 tiles on 64-k slices (round 5: the same finding for the one-round GEMMs of the `fast` precision — the decoder's 3200-row
// projections at the headline batch, everything at one image): 28 / 32 / 16 KiB per slice: five, five and seven whole slots.
constexpr bool gemm_k64_plain(int wm, int wn, int tm, int tn, int stages) {
  return (wm == 2 && wn == 2 && tm == 4 && tn == 3 && stages == 5)                        // 128 x 96, 4 waves of 64 x 48, 5 x 28 KiB
      || (wm == 2 && wn == 2 && tm == 4 && tn == 4 && stages == 5)                        // 128 x 128, 4 waves of 64 x 64, 5 x 32 KiB
      || (wm == 2 && wn == 2 && tm == 2 && tn == 2 && stages == 7);                       // 64 x 64, 4 waves of 32 x 32, 7 x 16 KiB
}

// WM x WN waves; each wave owns TM x TN subtiles of 16x16.  Block tile = (WM*TM*16) x (WN*TN*16).
// STAGES = depth of the LDS ring: 4 for the big tiles; 8 for the small-tile variants used when a GEMM has fewer tiles than
// the chip has CUs — those are bound by bytes in flight per CU (3 x 16 KiB per 128x128 block = 24 GB/s per CU at ~2 us of
// loaded latency), so the ring, not the tile, is what has to grow.
// OUT: 0 = f32, 1 = f16, 2 = split pair (hi plane at C, lo plane at C + planeC).  SPLIT: operands are split pairs.
template <int WM, int WN, int TM, int TN, int STAGES, int OUT, int ACT, int VEC, int SPLIT>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN) >= 8 ? 2 : ((WM * WN * TM * TN) >= 96 ? 2 : 1)) void gemm_f16_kernel(GemmArgs p) {
  constexpr int NW = WM * WN;
  constexpr int NPL = SPLIT ? 2 : 1;        // planes of A
  constexpr int NPLW = SPLIT == 1 ? 2 : 1;  // planes of W (SPLIT = 2: W is exactly its hi plane)
  constexpr int OUT_F16 = OUT == 1;
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  // K64 (round 4): split-pair tiles below the big ones on TWO slots stage slices of 64 k (128-B row pieces = whole cache lines) instead
  // of 32: a CU's LDS-DMA stream moves 85 GB/s in 128-B pieces against 52 - 57 in 64-B pieces whatever the ring depth or the number of
  // issuing waves (tools/micro/dma_stream.hip, the batch-1 QKV pattern), and with one image's ~1200 token rows the K loop IS that
  // stream (ablations: profiles/NOTES.md round 4).
  constexpr bool K64 = BM * BN < 192 * 256 && (SPLIT ? (STAGES == 2 || gemm_k64_deep(WM, WN, TM, TN, STAGES, SPLIT)) : gemm_k64_plain(WM, WN, TM, TN, STAGES));
  constexpr int KB = K64 ? 64 : BK;         // k per staged slice
  constexpr int RPP = 512 / KB;             // rows per 1-KiB DMA piece (16 at 64-B rows, 8 at 128-B rows)
  constexpr int LPR = 64 / RPP;             // lanes (16-byte chunks) per row
  constexpr int ROWS = NPL * BM + NPLW * BN;   // LDS rows per stage (A planes then W planes), 2 * KB bytes each
  constexpr int PIECES = ROWS / RPP;        // 1-KiB DMA pieces per stage
  constexpr int NP = (PIECES + NW - 1) / NW;  // DMA issues per wave per stage (duplicates pad uneven splits)
  static_assert(NP >= 2 && NP <= 8 && STAGES >= (SPLIT ? 2 : 3) && STAGES <= 8, "unsupported pieces-per-wave count / ring depth");
  // Prefetch distance: slice kt+STAGES-1 goes into the slot whose fragments were consumed before the current barrier.
  constexpr int DIST = STAGES - 1;
  constexpr int AHEAD = SPLIT ? 1 : DIST - 2; // whole stages that may still be in flight at a steady-state barrier
  static_assert(AHEAD >= 1 && AHEAD * NP < 64, "ring too shallow / vmcnt overflow");
  constexpr int STAGE_HALVES = ROWS * KB;
  // PERS (round 5): the big plain-fp16 tiles (8 waves, 4-slot ring) are PERSISTENT — a workgroup walks tiles vb, vb + gridDim.x, ... and
  // requests the next tile's first DIST slices right behind the barrier that opens this tile's epilogue, so the ~2 us of first-byte
  // latency (and the dispatch of a new workgroup) land under the epilogue's 3.5 - 13 us of stores instead of in front of the next K
  // loop (in-kernel probe, QKV 14144 x 2304 x 768: prologue 2.1 / K loop 19.0 / epilogue 3.5 us per tile).  The epilogue's slabs live
  // in the ring slot the prologue does not use plus the LDS beyond the ring (the kernel takes all 160 KiB), in smaller passes.
  // Same arithmetic per output element: results are bitwise those of one workgroup per tile (the launcher picks the grid).
  // (the fp32-output 256 x 256 form stays one workgroup per tile: with the tile loop around it hipcc spills 36 bytes in its epilogue)
  constexpr bool PERS = !SPLIT && NW == 8 && STAGES == 4 && VEC == 2 && (OUT == 1 || TN == 3);
  constexpr int SLAB_OFF = PERS ? (STAGES - 1) * STAGE_HALVES * 2 : 0;                       // bytes: slot STAGES - 1 and what follows
  constexpr int RINGP = K64 ? gemm_k64_ring_pieces(WM, WN, TM, TN, STAGES, SPLIT) : 0;            // circular ring of pieces (K64 tiles too large for three slots)
  constexpr bool FRAC = RINGP != 0;
  constexpr int LDS_BYTES = PERS ? 160 * 1024 : (FRAC ? RINGP * 1024 : STAGES * STAGE_HALVES * 2);
  constexpr int SLAB_CAP = LDS_BYTES - SLAB_OFF;
  __shared__ __attribute__((aligned(16))) half_t smem[LDS_BYTES / 2];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform values live in SGPRs
  const int wr = wave / WN, wc = wave % WN;
  ZH_PROBE(0);

  // XCD-aware bijective remap: blocks b, b+8, ... share an XCD -> give each XCD a contiguous id range.  `vb` is the virtual block
  // id: blockIdx.x for one workgroup per tile; blockIdx.x + i * gridDim.x for the i-th tile of a persistent workgroup (gridDim.x is
  // a multiple of 8 then: every tile of a workgroup keeps its XCD's id range, and at any moment an XCD runs consecutive ids)
  auto decode = [&](int vb, int& batch_, int& tm_, int& tn_) {
    const int nwg = PERS ? p.total : (int)gridDim.x;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int tiles = p.nbm * p.nbn;
    const int xcd = vb & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (vb >> 3);
    batch_ = wg / tiles;
    const int trem = wg - batch_ * tiles;
    // super-tile order: GROUP_M consecutive ids walk GROUP_M m-tiles of one n-tile
    const int gsz = p.group_m * p.nbn;
    const int gid = trem / gsz;
    const int gfirst = gid * p.group_m;
    const int grows = min(p.nbm - gfirst, p.group_m);
    const int gl = trem - gid * gsz;
#ifdef ZH_X_WALK_N                                      // developer A/B: consecutive ids walk the n-tiles of one m-tile
    tm_ = gfirst + gl / p.nbn; tn_ = gl % p.nbn;
#else
    tm_ = gfirst + gl % grows; tn_ = gl / grows;
#endif
  };
  int vb = blockIdx.x;
  int batch, tm, tn;
  decode(vb, batch, tm, tn);
  int m0 = tm * BM, n0 = tn * BN;

  // DMA sources: piece pc covers LDS rows [16*pc, 16*pc+16); lane -> row (lane>>2), phys chunk lane&3.
  // LDS row order: A hi [BM] (A lo [BM]) W hi [BN] (W lo [BN]).  A piece never straddles two row sets, so its source is a
  // wave-uniform base (SGPR pair, advanced by BK per slice) + a per-lane 32-bit element offset that never changes.
  const half_t* gbase[NP];
  unsigned goff[NP];
  int lds_piece[NP];
  auto setup_dma = [&](int b_, int m0_, int n0_, int lane) {   // the tile's operand rows -> gbase / goff (`lane`: see lane_e in the epilogue)
    const half_t* A = p.A + (long)b_ * p.sA;
    const half_t* W = p.W + (long)b_ * p.sW;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      int pc = wave + i * NW;
      pc = pc < PIECES ? pc : PIECES - 1;
      lds_piece[i] = pc * RPP * KB;
      const int R0 = pc * RPP;                      // uniform
      const int rl = lane / LPR;
      // 64-B rows: chunk ^ (-(row >> 2) & 3), (R0 + rl) >> 2 == R0/4 + (rl >> 2), R0/4 % 4 == 0.  128-B rows (K64): chunk ^ ((row >> 1) & 7)
      // — with it the sixteen lanes ds_read_b128 services together (rows 0-3, 12-15 of one k-chunk and 4-11 of its neighbour) fall into
      // sixteen different 16-byte bank groups: even rows take 0-7, odd rows 8-15, and (c ^ s) is a bijection over the eight rows of a parity
      const int c = K64 ? ((lane & 7) ^ (((R0 + rl) >> 1) & 7)) : ((lane & 3) ^ ((-(rl >> 2)) & 3));
      if (R0 < NPL * BM) {
        const int pl = SPLIT ? (R0 >= BM) : 0;
        int row = m0_ + (R0 - pl * BM) + rl;
        row = row < p.M ? row : p.M - 1;
        gbase[i] = A + pl * p.planeA;
        goff[i] = (unsigned)row * (unsigned)p.lda + c * 8;
        if constexpr (PERS) gbase[i] += goff[i];
      } else {
        const int Rw = R0 - NPL * BM;
        const int pl = SPLIT == 1 ? (Rw >= BN) : 0;
        int row = n0_ + (Rw - pl * BN) + rl;
        row = row < p.N ? row : p.N - 1;
        gbase[i] = W + pl * p.planeW;
        goff[i] = (unsigned)row * (unsigned)p.ldw + c * 8;
        if constexpr (PERS) gbase[i] += goff[i];
      }
    }
  };
  setup_dma(batch, m0, n0, lane);
  // PERS: ONE per-lane 64-bit source pointer per piece (base + lane offset folded at tile setup, advanced by a 64-bit add per issue)
  // instead of a uniform base + a 32-bit lane offset: hipcc forms 64-bit per-lane addresses for the builtin anyway and kept BOTH forms
  // alive across the K loop (the offsets for the tail loops) — 8 registers the tile loop does not have
  auto issue_stage = [&](int slot) {
    half_t* sb = smem + slot * STAGE_HALVES;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      if constexpr (PERS) __builtin_amdgcn_global_load_lds((glb_ptr_t)gbase[i], (lds_ptr_t)(sb + lds_piece[i]), 16, 0, 0);
      else __builtin_amdgcn_global_load_lds((glb_ptr_t)(gbase[i] + goff[i]), (lds_ptr_t)(sb + lds_piece[i]), 16, 0, 0);
      gbase[i] += KB;
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / KB;                    // K % 64 == 0: even at 32-k slices, any count >= 1 at 64
  const int frow = lane & 15, fk = lane >> 4;
  // ---- pos tables: accumulators start from pos_y[y] + pos_x[x] (N % 4 == 0 checked on the host).  SPLIT: the accumulator is
  // scaled by out_scale = 2^-s afterwards — start from the table value times 2^s (exact).
  // In the MFMA accumulator layout a lane owns 4 columns of ONE row per 16x16 subtile, so reading the tables straight from
  // global memory costs TM*TN*2 loads per lane that each touch 16 different table rows: ~1 MB of L1 line traffic per
  // 256x256 tile for a 25-50 KB unique slice (measured: +130 us on the 230-us K projection, fp16 or fp32 tables alike).
  // Instead the block copies the tile's slice — all pos_w rows of pos_x and the few rows of pos_y its pixels span, BN
  // columns — into the ring slot the prologue leaves free (slot STAGES-1), every line fetched once, and the lanes pick their
  // rows out of LDS.  The copy's global loads are issued BEFORE the prologue's DMA (vmcnt retires in order: the counted waits
  // of the ring can only over-wait) and stored after it.  Slices that do not fit the slot (wide images) take the direct path.
  constexpr int POS_SLOT_BYTES = STAGE_HALVES * 2, POS_NT = 64 * NW;
  // (SPLIT = 2: at most 2 passes — more register sets spilled in the 256 x 256 tiles; larger slices take the direct path.  No
  //  x2 GEMM of the model carries pos tables: the composed K / V weights that do are never fp16-valued.)
  constexpr int POS_MAXIT_FULL = (POS_SLOT_BYTES / 16 + POS_NT - 1) / POS_NT;
  constexpr int POS_MAXIT = SPLIT == 2 && POS_MAXIT_FULL > 2 ? (TN >= 8 ? 1 : 2) : POS_MAXIT_FULL;   // (TN = 8: one pass — a second register set spilled once the tile loop of round 5 wrapped the kernel)
  constexpr int POS_CAP_BYTES = POS_MAXIT * POS_NT * 16 < POS_SLOT_BYTES ? POS_MAXIT * POS_NT * 16 : POS_SLOT_BYTES;
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  u32x4_t pos_v[POS_MAXIT];
  const unsigned pos_q0 = p.pos_y ? (unsigned)m0 % (unsigned)p.pos_hw : 0u;
  const unsigned pos_y0 = p.pos_y ? pos_q0 / (unsigned)p.pos_w : 0u, pos_x0 = pos_q0 - pos_y0 * (unsigned)p.pos_w;
  const int pos_rsb = BN * (p.pos_f16 ? 2 : 4) + 16;                          // LDS row stride of the slice (bytes)
  const int pos_rows = p.pos_y ? p.pos_w + (int)((pos_x0 + BM - 1) / (unsigned)p.pos_w) + 1 : 0;
  const bool pos_lds = !FRAC && p.pos_y && pos_rows * pos_rsb <= POS_CAP_BYTES;   // (the circular ring leaves no slot free under its prologue: direct path)
  char* const pos_slot = (char*)(smem + (STAGES - 1) * STAGE_HALVES);
  auto pos_fetch_t = [&](auto tag) {                            // global -> registers, one 16-byte chunk per thread and pass
    typedef decltype(tag) T;
    constexpr int EPC = 16 / (int)sizeof(T), CPR = BN / EPC;    // elements per chunk, chunks per row
    const int total = pos_rows * CPR, hgt = p.pos_hw / p.pos_w;
#pragma unroll
    for (int it = 0; it < POS_MAXIT; ++it) {
      int c = tid + it * POS_NT;
      c = c < total ? c : total - 1;                            // branch-free: the surplus threads re-fetch the last chunk
      const int row = c / CPR, ch = c - row * CPR;
      const bool isx = row < p.pos_w;                           // selects, not branches: the loads of all passes batch
      const unsigned ry = (pos_y0 + (unsigned)(isx ? 0 : row - p.pos_w)) % (unsigned)hgt;
      const T* base = isx ? (const T*)p.pos_x : (const T*)p.pos_y;
      int n = n0 + ch * EPC;
      n = n < p.N ? n : 0;                                      // columns >= N are never stored
      pos_v[it] = *(const u32x4_t*)(base + (long)(isx ? (unsigned)row : ry) * p.ld_pos + n);
    }
  };
  auto pos_apply_t = [&](auto tag) {                            // registers -> LDS slice -> accumulators
    typedef decltype(tag) T;
    typedef T T4 __attribute__((ext_vector_type(4)));
    constexpr int EPC = 16 / (int)sizeof(T), CPR = BN / EPC;
    const int total = pos_rows * CPR;
#pragma unroll
    for (int it = 0; it < POS_MAXIT; ++it) {
      const int c = tid + it * POS_NT;
      const int row = c / CPR, ch = c - row * CPR;
      if (c < total) *(u32x4_t*)(pos_slot + row * pos_rsb + ch * 16) = pos_v[it];
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const float isc = SPLIT ? 1.0f / p.out_scale : 1.0f;
    const int rmax = p.M - 1 - m0;
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
      int r = (wr * TM + mt) * 16 + frow;
      r = r < rmax ? r : rmax;                                  // rows >= M are never stored
      const unsigned q = pos_x0 + (unsigned)r, jy = q / (unsigned)p.pos_w, x = q - jy * (unsigned)p.pos_w;
      const char* lx = pos_slot + x * pos_rsb + ((wc * TN) * 16 + fk * 4) * (int)sizeof(T);
      const char* ly = pos_slot + (p.pos_w + jy) * pos_rsb + ((wc * TN) * 16 + fk * 4) * (int)sizeof(T);
#pragma unroll
      for (int nt = 0; nt < TN; ++nt) {
        const T4 a = *(const T4*)(lx + nt * 16 * (int)sizeof(T)), b = *(const T4*)(ly + nt * 16 * (int)sizeof(T));
        acc[nt][mt] = (f32x4){(float)a[0] + (float)b[0], (float)a[1] + (float)b[1], (float)a[2] + (float)b[2], (float)a[3] + (float)b[3]} * isc;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the slice is consumed before the ring may reuse the slot
  };
  auto pos_direct_t = [&](auto tag) {                           // slice larger than the slot: straight from global memory
    typedef decltype(tag) T;                                    // float or half_t table elements
    typedef T T4 __attribute__((ext_vector_type(4)));
    const float isc = SPLIT ? 1.0f / p.out_scale : 1.0f;
    // rows of subtiles per batch: the largest DIVISOR of TM with G * TN <= 16 (<= 32 loads in flight).  (16 / TN is not a
    // divisor of TM = 6: the 192 x 256 tile then wrote accumulators 6 and 7 of 6 — found by test_gemm_x3_pos_tables_every_tile)
    constexpr int G = TM * TN <= 16 ? TM : (TM % 4 == 0 && 4 * TN <= 16 ? 4 : (TM % 3 == 0 && 3 * TN <= 16 ? 3 : (TM % 2 == 0 && 2 * TN <= 16 ? 2 : 1)));
    static_assert(TM % G == 0, "pos_direct_t: the batch height must divide TM");
#pragma unroll
    for (int g = 0; g < TM; g += G) {
      T4 ry[G][TN], rx[G][TN];
#pragma unroll
      for (int ml = 0; ml < G; ++ml) {
        const int m = m0 + (wr * TM + g + ml) * 16 + frow;
        const unsigned pix = (unsigned)(m < p.M ? m : p.M - 1) % (unsigned)p.pos_hw;
        const unsigned y = pix / (unsigned)p.pos_w, x = pix - y * (unsigned)p.pos_w;
        const T* ty = (const T*)p.pos_y + (long)y * p.ld_pos;
        const T* tx = (const T*)p.pos_x + (long)x * p.ld_pos;
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
          int n = n0 + (wc * TN + nt) * 16 + fk * 4;
          n = n < p.N ? n : 0;                                 // branch-free (columns >= N are never stored): the loads batch
          ry[ml][nt] = *(const T4*)(ty + n);
          rx[ml][nt] = *(const T4*)(tx + n);
        }
      }
#pragma unroll
      for (int ml = 0; ml < G; ++ml)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
          const T4 a = ry[ml][nt], b = rx[ml][nt];
          acc[nt][g + ml] = (f32x4){(float)a[0] + (float)b[0], (float)a[1] + (float)b[1], (float)a[2] + (float)b[2], (float)a[3] + (float)b[3]} * isc;
        }
    }
  };
  auto pos_before_prologue = [&]() {
    if (!pos_lds) return;
    if (p.pos_f16) pos_fetch_t(half_t{});
    else pos_fetch_t(float{});
  };
  auto pos_after_prologue = [&]() {
    if (!p.pos_y) return;
    if (pos_lds) {
      if (p.pos_f16) pos_apply_t(half_t{});
      else pos_apply_t(float{});
    } else {
      if (p.pos_f16) pos_direct_t(half_t{});
      else pos_direct_t(float{});
    }
  };
  const int foff = K64 ? frow * KB : frow * BK + ((fk ^ ((-(frow >> 2)) & 3)) * 8);   // per-lane offset inside a 16-row subtile (K64: + the k-step's chunk)
  const half_t* rdA = smem + (wr * TM * 16) * KB + foff;
  const half_t* rdW = smem + (NPL * BM + wc * TN * 16) * KB + foff;

  if constexpr (PERS) {                       // the first tile's prologue, in FRONT of the tile loop: with the pos-table code inside it, everything
    pos_before_prologue();                    // in there that depends on the lane alone was hoisted out and kept alive through the K loop (spills)
#pragma unroll
    for (int s = 0; s < DIST; ++s)
      if (s < nk) issue_stage(s % STAGES);
    pos_after_prologue();
  }
  for (;;) {                                  // tiles of this workgroup: one, or (PERS) vb, vb + gridDim.x, ...
  const int vb_next = vb + (int)gridDim.x;
  const bool more = PERS && vb_next < p.total;
#ifdef ZH_ABL_TAIL_SPLIT      // developer ablation (timing only, results are garbage; round 6): what an IDEAL stream-K / fixed-split tail round could
  // return — the `rem` tiles of the last, partial round of a persistent launch run rem / gridDim.x of their K slices each, i.e. the round
  // lasts as long as if its work were spread evenly over every workgroup, with no reduction traffic at all
  int nk_tile = nk;
  if (PERS && (int)gridDim.x < p.total) {
    const int rem = p.total % (int)gridDim.x;
    if (rem && vb >= p.total - rem) nk_tile = max(2 * DIST, ((nk * rem / (int)gridDim.x) + 1) & ~1);
  }
#define nk nk_tile
#endif

  constexpr bool BIGT = SPLIT && BM * BN >= 192 * 256;    // the two-slot (SPLIT = 2: three-slot) big tiles
  static_assert(!SPLIT || K64 || BIGT == (STAGES == 2 || (SPLIT == 2 && STAGES == 3 && BM * BN >= 192 * 256)), "big split-pair tiles: 2 slots (x2: 2 or 3)");
  if constexpr (K64) {
    // ---- f16x3 loop on 64-k slices.  Per slice: counted wait + barrier (slice kt landed everywhere, every read of the slot about to be
    // refilled has returned) -> fragments of both k-steps -> DMA of slice kt + STAGES - 1 into the slot slice kt - 1 left -> 2 x three sweeps.
    // Two slots: one slice of prefetch (the round-4 form).  Three and more: STAGES - 2 slices stay in flight across the barrier.
    constexpr int KD = STAGES - 1;
    const int sz = (frow >> 1) & 7;
    if constexpr (FRAC) {
    // ---- circular ring.  Stream position of piece pc of slice s: s * PIECES + pc; ring position: that modulo RINGP (both multiples of 8
    // pieces = 64 rows: a 16-row fragment read never straddles the wrap, and wave w always owns the ring positions = w mod NW).  A slice
    // goes out in two parts: its first H issues per wave two steps ahead, the other NP - H one step ahead, so that behind barrier kt exactly
    // one slice's worth of pieces — the rest of slice kt + 1, then the head of slice kt + 2 — refills the space slice kt - 1 left.
    // At barrier kt slice kt must have landed: the only issues younger than its last piece are the H of slice kt + 1's head.
    static_assert(PIECES % NW == 0 && RINGP % 8 == 0 && PIECES % 8 == 0 && RPP == 8, "circular ring: whole issues per wave, wrap on 64-row boundaries");
    constexpr int H = (RINGP - 2 * PIECES) / NW;
    static_assert((RINGP - 2 * PIECES) % NW == 0 && H >= 1 && H < NP && NP + H < 64, "circular ring: between two and three slices");
    auto wrap = [](int r) { return r >= RINGP ? r - RINGP : r; };
    auto issue_part = [&](int st, auto LO, auto HI) {          // issues [LO, HI) of the slice whose ring start is piece st
#pragma unroll
      for (int i = decltype(LO)::value; i < decltype(HI)::value; ++i) {
        const int r = wrap(st + wave + i * NW);
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(gbase[i] + goff[i]), (lds_ptr_t)(smem + r * 512), 16, 0, 0);
        gbase[i] += KB;
      }
    };
    typedef std::integral_constant<int, 0> i0_t;
    typedef std::integral_constant<int, H> iH_t;
    typedef std::integral_constant<int, NP> iN_t;
    pos_before_prologue();
    if (nk > 0) issue_part(0, i0_t{}, iN_t{});
    if (nk > 1) issue_part(PIECES, i0_t{}, iN_t{});
    if (nk > 2) issue_part(2 * PIECES, i0_t{}, iH_t{});
    pos_after_prologue();
    ZH_PROBE(1);
    half8_t fa[2][2 * TM], fw[2][NPLW * TN];
    const int lofs0 = foff + ((fk ^ sz) * 8), lofs1 = foff + (((4 | fk) ^ sz) * 8);   // lane offsets (halves) inside a subtile's two pieces, k-steps 0 / 1
    int st = 0;                                                // ring start (piece) of slice kt
    for (int kt = 0; kt < nk; ++kt) {
      if (kt == 0) {
        if (nk > 2) wait_vmcnt_barrier<NP + H>();
        else if (nk > 1) wait_vmcnt_barrier<NP>();
        else wait_vmcnt_barrier<0>();
      } else if (kt + 1 < nk) wait_vmcnt_barrier<H>();
      else wait_vmcnt_barrier<0>();
      // the refill goes out FIRST: the tail of slice kt + 1 is needed one step from now, and a step (fragment reads + 6 TM TN MFMAs) is about
      // one load latency long
      const int st1 = wrap(st + PIECES);
      if (kt >= 1) {
        if (kt + 1 < nk) issue_part(st1, iH_t{}, iN_t{});
        if (kt + 2 < nk) issue_part(wrap(st1 + PIECES), i0_t{}, iH_t{});
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int lo = j ? lofs1 : lofs0;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
          for (int t = 0; t < TM; ++t) fa[j][pl * TM + t] = *(const half8_t*)(smem + wrap(st + pl * (BM / 8) + (wr * TM + t) * 2) * 512 + lo);
          if (pl < NPLW) {
#pragma unroll
            for (int t = 0; t < TN; ++t) fw[j][pl * TN + t] = *(const half8_t*)(smem + wrap(st + (NPL * BM + pl * BN) / 8 + (wc * TN + t) * 2) * 512 + lo);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int sw = 0; sw < 3; ++sw) {
          if (SPLIT == 2 && sw == 1) continue;           // W has no lo plane
#pragma unroll
          for (int nt = 0; nt < TN; ++nt)
#pragma unroll
            for (int mt = 0; mt < TM; ++mt)
              acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[j][(sw == 1 ? TN : 0) + nt], fa[j][(sw == 2 ? TM : 0) + mt], acc[nt][mt], 0, 0, 0);
        }
      st = st1;
    }
    } else {
    static_assert((KD - 1) * NP < 64, "vmcnt overflow");
    pos_before_prologue();
#pragma unroll
    for (int s = 0; s < KD; ++s)
      if (s < nk) issue_stage(s);
    pos_after_prologue();
    ZH_PROBE(1);
    half8_t fa[2][NPL * TM], fw[2][NPLW * TN];
    int slot = 0, wslot = KD % STAGES;
    auto read_frags = [&]() {
      const int so = slot * STAGE_HALVES;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ko = (((j << 2) | fk) ^ sz) * 8;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
          for (int t = 0; t < TM; ++t) fa[j][pl * TM + t] = *(const half8_t*)(rdA + so + (pl * BM + t * 16) * KB + ko);
          if (pl < NPLW) {
#pragma unroll
            for (int t = 0; t < TN; ++t) fw[j][pl * TN + t] = *(const half8_t*)(rdW + so + (pl * BN + t * 16) * KB + ko);
          }
        }
      }
    };
    auto sweeps = [&]() {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int sw = 0; sw < (SPLIT ? 3 : 1); ++sw) {   // (plain fp16 operands: the one product)
          if (SPLIT == 2 && sw == 1) continue;           // W has no lo plane
#pragma unroll
          for (int nt = 0; nt < TN; ++nt)
#pragma unroll
            for (int mt = 0; mt < TM; ++mt)
              acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[j][(sw == 1 ? TN : 0) + nt], fa[j][(sw == 2 ? TM : 0) + mt], acc[nt][mt], 0, 0, 0);
        }
    };
    auto advance = [&]() {
      slot = slot + 1 == STAGES ? 0 : slot + 1;
      wslot = wslot + 1 == STAGES ? 0 : wslot + 1;
    };
    int kt = 0;
    for (; kt + KD < nk; ++kt) {
      wait_vmcnt_barrier<(KD - 1) * NP>();
      read_frags();
      issue_stage(wslot);
      sweeps();
      advance();
    }
    for (; kt < nk; ++kt) {              // tail: slices kt .. nk-1 are in flight, slice kt must have landed
      wait_stages_barrier<NP, KD - 1>(nk - 1 - kt);
      read_frags();
      sweeps();
      advance();
    }
    }
  } else if constexpr (BIGT) {
    // ---- f16x3 loop, big tile (256 x 256, 8 waves of 128 x 64), TWO 64-KiB slots.  Ablations of the 3-slot 256 x 128 loop
    // (tools/gemm_x3_probe.sh, round 3; QKV shape, model-shaped operands): all 160 us; MFMAs removed 116 us; operand
    // movement removed (no DMA, no fragment reads) 119 us — the LDS-DMA stream (48 KiB per slice and CU, at its request-rate
    // limit) is as long a leg as the MFMAs, and the two overlap poorly.  A 256 x 256 tile moves 64 KiB per slice for TWICE
    // the MFMAs (-33 % bytes and -25 % fragment reads per MFMA).  Its ring has room for two slots only, but a slice now
    // carries 96 MFMAs per wave (~1.5 us at two waves per SIMD): one slice of prefetch covers the load latency.
    // Per slice: barrier (slice kt landed everywhere, everyone is done with the other slot) -> DMA of slice kt+1 into the other
    // slot -> fragments hi planes first -> three sweeps (hi*hi, lo_w*hi_a, hi_w*lo_a).
    // LDS-DMA issue with a wave-uniform 64-bit base in SGPRs and a constant 32-bit per-lane BYTE offset (the `saddr` form of
    // global_load_lds, from inline asm): the builtin's flat-pointer form made hipcc keep eight 64-bit per-lane addresses in
    // VGPRs and re-derive them with 64-bit VALU adds per issue — 16+ registers this tile does not have (it spilled).
    const half_t* sbase[NP];
    unsigned goffb[NP], ldsb[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      // wave-uniform by construction; readfirstlane makes it so for the compiler too (the "s" constraint below was handed a
      // VGPR pair — an assembler error — in the -DZH_GEMM_PROBE build, where its uniformity analysis gave up)
      const uintptr_t gb = (uintptr_t)gbase[i];
      sbase[i] = (const half_t*)(((uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(gb >> 32)) << 32) |
                                 (uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)gb));
      goffb[i] = goff[i] * 2u;                                   // host check: operands below 2^31 elements for this tile
      ldsb[i] = (unsigned)(uintptr_t)smem + (unsigned)lds_piece[i] * 2u;
    }
    auto issue2 = [&](int slot) {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                     :: "s"(ldsb[i] + (unsigned)(slot * STAGE_HALVES * 2)), "v"(goffb[i]), "s"(sbase[i]) : "memory");
        sbase[i] += BK;
      }
    };
    pos_before_prologue();
    // Three slots: the counted wait of the loop (one stage in flight) only works if the compiler has no vector-memory load of its
    // own outstanding there.  The pos prefetch registers are consumed under a condition it cannot correlate with the one they
    // were loaded under, so it kept an `s_waitcnt vmcnt(0)` INSIDE the loop, on the first reuse of such a register — which also
    // drains every LDS-DMA (they are inline asm, invisible to its counter model) and turns two slices of prefetch into none.
    // A wait it can see, before the first DMA, retires them here (nothing is outstanding unless the GEMM carries pos tables).
    if constexpr (STAGES == 3) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched
    issue2(0);
    pos_after_prologue();
    ZH_PROBE(1);
    // Registers: 128 accumulators + A hi (32) + W hi (16) + W lo (16); a fourth fragment set does not fit (the build fails on
    // scratch).  The A lo fragments therefore REPLACE the A hi ones: the second sweep walks the A fragments in order and, as
    // soon as fragment mt has fed its TN MFMAs, its lo plane is read into the same registers — TM - 1 groups of MFMAs ahead
    // of the third sweep's first use.
    // SPLIT = 2 (W has no lo plane): two sweeps, see the loop body; a slot is 48 KiB, so the ring may hold THREE (two slices of
    // prefetch: a slice is only 64 MFMAs per wave, ~1.2 us).
    constexpr int H1 = (TM + 1) / 2, H2 = TM - H1;     // SPLIT = 2: A lo fragments with registers of their own / replacing hi fragments
    half8_t fa[TM], fw[NPLW * TN], fl[SPLIT == 2 ? H1 : 1];
    auto read_hi = [&](int slot) {
      const int so = slot * STAGE_HALVES;
#pragma unroll
      for (int t = 0; t < TM; ++t) fa[t] = *(const half8_t*)(rdA + so + (t * 16) * BK);
#pragma unroll
      for (int t = 0; t < TN; ++t) fw[t] = *(const half8_t*)(rdW + so + (t * 16) * BK);
      if constexpr (SPLIT == 1) {
#pragma unroll
        for (int t = 0; t < TN; ++t) fw[TN + t] = *(const half8_t*)(rdW + so + (BN + t * 16) * BK);
      } else {
#pragma unroll
        for (int t = 0; t < H1; ++t) fl[t] = *(const half8_t*)(rdA + so + (BM + t * 16) * BK);
      }
    };
    // (Measured and not kept, round 3: the two waves of a SIMD taking their DMA issues at different points of the slice — waves
    //  4 .. 7 after the hi * hi sweep — so that one feeds the MFMA pipe while the other sits in its ~1000 cycles of issue stalls:
    //  QKV 147 -> 153 us, c_fc 189 -> 195, K / V 479 -> 488.  The later issue costs the landing time it was meant to hide.)
    // `inflight` = whole stages that may still be in flight at the barrier (0: two slots, 1: the steady state of three)
    auto body = [&](int slot, int nslot, bool prefetch, auto inflight) {
      const int so = slot * STAGE_HALVES;
#ifndef ZH_X3_NOBAR
      // lgkmcnt(0) too: hipcc moves the last MFMAs of the previous slice (register-only) below this point and with them the
      // wait for the fragment reads they consume — every read of the slot the DMA below overwrites must have RETURNED first
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(decltype(inflight)::value * NP) : "memory");
#endif
#ifndef ZH_X3_NOFRAG
      read_hi(slot);
#endif
#ifndef ZH_X3_NODMA
      if (prefetch) issue2(nslot);
#endif
#ifndef ZH_X3_NOMFMA
#pragma unroll
      for (int nt = 0; nt < TN; ++nt)                  // hi_w * hi_a
#pragma unroll
        for (int mt = 0; mt < TM; ++mt)
          acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[nt], fa[mt], acc[nt][mt], 0, 0, 0);
      if constexpr (SPLIT == 1) {
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {                // lo_w * hi_a, then this A fragment's lo plane takes its place
#pragma unroll
          for (int nt = 0; nt < TN; ++nt)
            acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[TN + nt], fa[mt], acc[nt][mt], 0, 0, 0);
#ifndef ZH_X3_NOFRAG
          fa[mt] = *(const half8_t*)(rdA + so + (BM + mt * 16) * BK);
#endif
        }
#pragma unroll
        for (int mt = 0; mt < TM; ++mt)                  // hi_w * lo_a
#pragma unroll
          for (int nt = 0; nt < TN; ++nt)
            acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[nt], fa[mt], acc[nt][mt], 0, 0, 0);
      } else {
        // no lo_w sweep to read the A lo fragments under: the first half of them has a register set of its own (the 16 registers
        // the W lo fragments would take), requested with the hi fragments and landing under the hi * hi sweep; the second half
        // replaces the first hi fragments as soon as that sweep has issued and lands under the first half's MFMAs.  (Left to
        // itself hipcc fused the sweeps per A fragment — 2 reads, wait, 8 MFMAs, 8 times per slice: every LDS latency exposed.)
#ifndef ZH_X3_NOFRAG
#pragma unroll
        for (int t = 0; t < H2; ++t) fa[t] = *(const half8_t*)(rdA + so + (BM + (H1 + t) * 16) * BK);
#endif
#pragma unroll
        for (int mt = 0; mt < H1; ++mt)                // hi_w * lo_a, rows of the first half
#pragma unroll
          for (int nt = 0; nt < TN; ++nt)
            acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[nt], fl[mt], acc[nt][mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < H2; ++mt)                // hi_w * lo_a, rows of the second half
#pragma unroll
          for (int nt = 0; nt < TN; ++nt)
            acc[nt][H1 + mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[nt], fa[mt], acc[nt][H1 + mt], 0, 0, 0);
      }
#else
      acc[0][0] += (f32x4){(float)fa[0][0], (float)fw[0][1], (float)fa[TM - 1][2], (float)fw[(NPLW - 1) * TN][3]};
#endif
    };
#ifdef ZH_X3_NOFRAG
    read_hi(0);
#endif
    typedef std::integral_constant<int, 0> fl0_t;
    typedef std::integral_constant<int, 1> fl1_t;
    if constexpr (STAGES == 2) {
      int kt = 0;
      for (; kt + 2 < nk; kt += 2) {       // nk is even: two slices per trip keep the slot index a compile-time constant
        body(0, 1, true, fl0_t{});
        body(1, 0, true, fl0_t{});
      }
      body(0, 1, true, fl0_t{});
      body(1, 0, false, fl0_t{});
    } else {
      // three slots: slices kt and kt + 1 are in flight when body(kt) starts; slice kt + 2 goes into the slot body(kt - 1) read,
      // which every wave has left once it passes body(kt)'s barrier.  The slot index is a run-time scalar here (two scalar
      // adds per slice); the steady loop is branch-free.
      issue2(1);                           // nk >= 2
      int slot = 0, nslot = 2;
      int kt = 0;
      for (; kt + 2 < nk; ++kt) {
        body(slot, nslot, true, fl1_t{});
        slot = slot == 2 ? 0 : slot + 1;
        nslot = nslot == 2 ? 0 : nslot + 1;
      }
      body(slot, 0, false, fl1_t{});
      body(slot == 2 ? 0 : slot + 1, 0, false, fl0_t{});
    }
  } else if constexpr (SPLIT) {
    // ---- f16x3 loop.  Fragments are single-buffered (hi + lo of both operands = 64 registers at 64x64 per wave; a second
    // set does not fit next to the accumulators at two waves per SIMD): slice kt's fragments are read right after the
    // barrier of iteration kt, hi planes first, and the first sweep of MFMAs (hi*hi) starts as soon as those arrive while
    // the lo planes and the DMA issues of slice kt+DIST trickle in underneath.  A slice carries 3*TM*TN MFMAs (~770 cycles
    // per wave), so two slices of prefetch cover ~1.5 us of load latency with a 3-slot ring.
    // (Measured and not kept, round 3: register double-buffered fragments for the 64 x 64 tile — slice kt + 1's reads and the DMA issue
    //  under slice kt's MFMAs, bit-identical — config-3 forward 3.00 vs 2.98 ms, batch-1 336 px 2.51 vs 2.50: those few-tile GEMMs
    //  cost ~7.5 us of launch + first-bytes + epilogue and only ~0.24 us per K slice; the loop is not what bounds them.)
    constexpr int DISTX = STAGES - 1, AHEADX = STAGES - 2;
    static_assert(AHEADX >= 1 && AHEADX * NP < 64, "ring too shallow / vmcnt overflow");
    pos_before_prologue();
#pragma unroll
    for (int s = 0; s < DISTX; ++s)
      if (s < nk) issue_stage(s);
    pos_after_prologue();
    ZH_PROBE(1);
    half8_t fa[2 * TM], fw[NPLW * TN];
    int slot = 0, wslot = DISTX % STAGES;
    auto read_frags = [&]() {
      const int so = slot * STAGE_HALVES;
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
        for (int t = 0; t < TM; ++t) fa[pl * TM + t] = *(const half8_t*)(rdA + so + (pl * BM + t * 16) * BK);
        if (pl < NPLW) {
#pragma unroll
          for (int t = 0; t < TN; ++t) fw[pl * TN + t] = *(const half8_t*)(rdW + so + (pl * BN + t * 16) * BK);
        }
      }
    };
    // three sweeps over the accumulators keep dependent MFMAs TM*TN issues apart: hi*hi, lo_w*hi_a, hi_w*lo_a
    auto sweeps = [&]() {
#pragma unroll
      for (int sw = 0; sw < 3; ++sw) {
        if (SPLIT == 2 && sw == 1) continue;           // W has no lo plane
#pragma unroll
        for (int nt = 0; nt < TN; ++nt)
#pragma unroll
          for (int mt = 0; mt < TM; ++mt)
            acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[(sw == 1 ? TN : 0) + nt], fa[(sw == 2 ? TM : 0) + mt], acc[nt][mt], 0, 0, 0);
      }
    };
    auto advance = [&]() {
      slot = slot + 1 == STAGES ? 0 : slot + 1;
      wslot = wslot + 1 == STAGES ? 0 : wslot + 1;
    };
    int kt = 0;
#ifdef ZH_X3_NOFRAG                      // developer ablations (tools/gemm_x3_probe.sh): timing only, results are garbage
    read_frags();
#endif
    for (; kt + DISTX < nk; ++kt) {      // steady: branch-free body so the reads / DMA issues interleave with the MFMAs
#ifndef ZH_X3_NOBAR
      wait_vmcnt_barrier<AHEADX * NP>();
#endif
#ifdef ZH_X3_DMA_FIRST                   // developer A/B (round 4): the next slice's DMA in front of this slice's fragment reads
      issue_stage(wslot);
      read_frags();
#else
#ifndef ZH_X3_NOFRAG
      read_frags();
#endif
#ifndef ZH_X3_NODMA
      issue_stage(wslot);
#endif
#endif
#ifndef ZH_X3_NOMFMA
      sweeps();
#else
      acc[0][0] += (f32x4){(float)fa[0][0], (float)fw[0][1], (float)fa[TM][2], (float)fw[(NPLW - 1) * TN][3]};
#endif
      // hi fragments first, then 2 MFMAs per lo-fragment read / DMA issue
      __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
      for (int i = 0; i < TM + (NPLW - 1) * TN; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      advance();
    }
    for (; kt < nk; ++kt) {              // tail: slices kt .. nk-1 are in flight, slice kt must have landed
      wait_stages_barrier<NP, AHEADX>(nk - 1 - kt);
      read_frags();
      sweeps();
      __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
      for (int i = 0; i < TM + (NPLW - 1) * TN; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      advance();
    }
  } else {
  if constexpr (!PERS) {                      // (PERS: requested in front of the tile loop / under the previous tile's epilogue)
    pos_before_prologue();
#pragma unroll
    for (int s = 0; s < DIST; ++s)
      if (s < nk) issue_stage(s % STAGES);
    pos_after_prologue();
  }

  // Register double-buffered fragments: while the MFMAs of slice kt run, the ds_read_b128 of slice kt+1 are in
  // flight (the LDS latency at the head of every slice was exposed on all 8 waves at once behind the barrier).
  // Slice kt+1 must therefore have landed one iteration earlier: counted waits are vmcnt(AHEAD*NP) in the steady state.
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
  // steady-state phase (kt + DIST < nk): branch-free so the scheduler can interleave — every group of MFMAs shadows one
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
    issue_stage((kt + DIST) % STAGES);
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
      // slices issued so far: 0 .. min(nk-1, kt+DIST-1); slice kt+1 must have landed, the later ones may fly
      wait_stages_barrier<NP, AHEAD>(min(nk - 1, kt + DIST - 1) - (kt + 1));
      load_frags(kt + 1, na, nw);                                          // reads before the DMA: see steady()
      if (kt + DIST < nk) issue_stage((kt + DIST) % STAGES);
    }
    mfma_all(fa, fw);   // (a sched_group_barrier interleave here makes hipcc spill: 172 scratch ops, 2.6x slower)
  };
  if (nk >= DIST) wait_vmcnt_barrier<(DIST - 1) * NP>();   // stage 0 landed; the other DIST-1 may still be in flight
  else wait_vmcnt_barrier<0>();                            // short K: not worth a counted wait
  ZH_PROBE(1);
  load_frags(0, fa0, fw0);
  int kt = 0;
  for (; kt + DIST + 1 < nk; kt += 2) {
    steady(kt, fa0, fw0, fa1, fw1);
    steady(kt + 1, fa1, fw1, fa0, fw0);
  }
  for (; kt < nk; kt += 2) {
    tail(kt, fa0, fw0, fa1, fw1);
    tail(kt + 1, fa1, fw1, fa0, fw0);
  }
  }

#ifdef ZH_ABL_TAIL_SPLIT
#undef nk
#endif
  ZH_PROBE(2);
  // ---- epilogue: lane owns rows m = ..+(lane&15), 4 consecutive n at 4*(lane>>4).  ACT / VEC are template
  // parameters: a runtime switch unrolled 32x blew the instruction cache (fc GEMM 1.4x slower in the model).
  // PERS: the epilogue sits inside the tile loop, and everything in it that depends on the lane alone is loop-invariant — hoisted in
  // front of the loop it would stay live through the K loop, which has no register to spare (the build spilled 300+ bytes).  An opaque
  // copy of the lane id pins those computations here.
  int lane_e = lane;
  if (PERS) asm volatile("" : "+v"(lane_e));
  const int frow_e = PERS ? (lane_e & 15) : frow, fk_e = PERS ? (lane_e >> 4) : fk;
  const long cb = (long)batch * p.sC;
  const float* R = p.R ? p.R + (long)batch * p.sR : nullptr;
  const float osc = SPLIT ? p.out_scale : 1.0f;
  const bool res_nowrap = p.res_rows >= p.M;               // the residual has its own row for every output row: no modulo
  if (VEC == 2) {
    // LDS-staged epilogue: the direct form stores 32-byte runs (4 lanes x 8 B) into 16 different 128-B lines per
    // instruction and measured 2.4 TB/s, fully exposed (34 % of a K=768 tile).  Here each wave transposes its tile through
    // a private, conflict-free LDS slab (row stride +16 B) and writes whole rows with 16 B per lane (split pairs: the
    // slab holds fp32 and each lane writes 8 B to the hi plane and 8 B to the lo plane).
    constexpr int ESZ = OUT_F16 ? 2 : 4;
    constexpr int RS = TN * 16 * ESZ + 16;                  // slab row stride (bytes)
    constexpr int PRW = (OUT_F16 ? 64 : 32) / (TN >= 8 ? 2 : 1);   // wide wave tiles: half the rows per pass (the pass's residual / slab registers)
    constexpr int PR0 = PRW < TM * 16 ? PRW : TM * 16;
    constexpr int PR1 = (TM * 16) % PR0 == 0 ? PR0 : (TM * 16 <= 48 ? TM * 16 : 16);      // rows per pass (divides the wave tile)
    // PERS: the slabs share what the next tile's prologue leaves of the LDS — halve the pass until they fit
    constexpr int PR = !PERS ? PR1 : (NW * PR1 * RS <= SLAB_CAP ? PR1 : (NW * (PR1 / 2) * RS <= SLAB_CAP ? PR1 / 2 : PR1 / 4));
    static_assert(PR >= 16 && PR % 16 == 0 && (TM * 16) % PR == 0 && NW * PR * RS <= SLAB_CAP, "epilogue slabs do not fit beside the next tile's prologue (a pass is whole 16-row sub-tiles)");
    constexpr int MTP = PR / 16;
    constexpr int CPRW = TN * 16 * ESZ / 16;                // 16-B chunks per row
    constexpr int NIT = PR * CPRW / 64;
    static_assert((PR * CPRW) % 64 == 0, "epilogue slab must divide into full wave reads");
    static_assert(SLAB_OFF + NW * PR * RS <= (int)sizeof(smem), "epilogue slabs exceed the LDS");
    // the TN bias vectors of this lane's columns, requested together and ONCE (they used to be loaded per pass and sub-tile
    // column behind a branch, each followed by a full wait: 16 exposed load latencies in the epilogue of a 256 x 256 tile)
    // (plain fp16 kernel, fp32 output, no activation — the residual GEMMs: the bias joins at store time instead, where a lane's
    //  chunk column never changes: ONE vector per lane for the whole epilogue instead of TN — (acc + bias) + residual either way)
    constexpr bool BIAS_LATE = !SPLIT && OUT == 0 && ACT == ZH_ACT_NONE && (64 % CPRW) == 0;
    f32x4 bvs[TN];
    f32x4 bias_late = {0.f, 0.f, 0.f, 0.f};
    if (BIAS_LATE) {
#pragma clang loop unroll(full)
      for (int nt = 0; nt < TN; ++nt) bvs[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (p.bias) {
        int n = n0 + wc * TN * 16 + (lane_e % CPRW) * (16 / ESZ);
        n = n < p.N ? n : 0;
        bias_late = *(const f32x4*)(p.bias + n);
      }
    } else
    if (p.bias) {                                           // wave-uniform
      const __attribute__((address_space(1))) float* gb = (const __attribute__((address_space(1))) float*)p.bias;
#pragma clang loop unroll(full)
      for (int nt = 0; nt < TN; ++nt) {
        int n = n0 + (wc * TN + nt) * 16 + fk_e * 4;
        n = n < p.N ? n : 0;                                // columns >= N are never stored
        bvs[nt] = *(const __attribute__((address_space(1))) f32x4*)(gb + n);
      }
    } else {
#pragma clang loop unroll(full)
      for (int nt = 0; nt < TN; ++nt) bvs[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();                                        // ring no longer read; every LDS-DMA has landed
    if (more) {                                             // PERS: the next tile's first slices fly under this epilogue
      int nb_, ntm_, ntn_;
      decode(vb_next, nb_, ntm_, ntn_);
      setup_dma(nb_, ntm_ * BM, ntn_ * BN, lane_e);
#pragma unroll
      for (int s = 0; s < DIST; ++s)
        if (s < nk) issue_stage(s % STAGES);
    }
    char* slab = (char*)smem + SLAB_OFF + wave * (PR * RS);
#pragma clang loop unroll(full)
    for (int pass = 0; pass < TM / MTP; ++pass) {
#ifdef ZH_ABL_SKIP_EPI        // developer ablation (timing only, results are garbage): a persistent tile that has a successor skips its slab passes and stores
      if (more) break;
#endif
#pragma clang loop unroll(full)
      for (int nt = 0; nt < TN; ++nt) {
        const f32x4 bv = bvs[nt];
#pragma clang loop unroll(full)
        for (int ml = 0; ml < MTP; ++ml) {
          f32x4 v = SPLIT ? acc[nt][pass * MTP + ml] * osc + bv : (BIAS_LATE ? acc[nt][pass * MTP + ml] : acc[nt][pass * MTP + ml] + bv);
          if (ACT != ZH_ACT_NONE) {
            v[0] = zh_act(v[0], ACT); v[1] = zh_act(v[1], ACT); v[2] = zh_act(v[2], ACT); v[3] = zh_act(v[3], ACT);
          }
          char* dst = slab + (ml * 16 + frow_e) * RS + (nt * 16 + fk_e * 4) * ESZ;
          if (OUT_F16) {
            half4_t h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            *(half4_t*)dst = h;
          } else {
            *(f32x4*)dst = v;
          }
        }
      }
      if constexpr (OUT == 2) {
        // split pair: a lane takes 8 consecutive columns — one 16-byte store per plane instead of two 8-byte ones (the epilogue
        // of a 256 x 256 tile was 7.9 us of a 29.6-us K = 256 block, store-issue bound: tools/gemm_x3_stamp.py)
        constexpr int UPR = CPRW / 2, NIT2 = PR * UPR / 64;
        static_assert(CPRW % 2 == 0 && (PR * UPR) % 64 == 0, "split-pair epilogue: 8-column units must tile the pass");
        // the residual (row-periodic table) of the whole pass is requested FIRST, branch-free (clamped addresses), so its loads
        // are in flight together — see the fp32 form below
        f32x4 rv0[NIT2], rv1[NIT2];
        if (R) {
#pragma clang loop unroll(full)
          for (int it = 0; it < NIT2; ++it) {
            const int c = it * 64 + lane_e;
            const int row = c / UPR, un = c - row * UPR;
            const int m = min(m0 + wr * TM * 16 + pass * PR + row, p.M - 1);
            int n = n0 + wc * TN * 16 + un * 8;
            n = n < p.N ? n : 0;
            const float* rp = R + (long)(res_nowrap ? m : m % p.res_rows) * p.ldr + n;
            rv0[it] = *(const f32x4*)rp; rv1[it] = *(const f32x4*)(rp + 4);
          }
        }
#pragma clang loop unroll(full)
        for (int it = 0; it < NIT2; ++it) {
          const int c = it * 64 + lane_e;
          const int row = c / UPR, un = c - row * UPR;
          const int m = m0 + wr * TM * 16 + pass * PR + row;
          const int n = n0 + wc * TN * 16 + un * 8;
          f32x4 d0 = *(const f32x4*)(slab + row * RS + un * 32), d1 = *(const f32x4*)(slab + row * RS + un * 32 + 16);
          if (m < p.M && n < p.N) {                       // N % 8 == 0 (host: wide_ok)
            if (R) { d0 += rv0[it]; d1 += rv1[it]; }
            zh_store_h8((half_t*)p.C + cb + (long)m * p.ldc + n, p.planeC, d0, d1);
          }
        }
      } else {
        // fp32 output + residual (out_proj, c_proj, the decoder's output projections): the loop used to load each residual
        // chunk right where it is added — read slab, ~30 address instructions (an integer modulo among them), ONE load, wait, add,
        // store — 24 exposed memory latencies per lane in a row: that, not bandwidth, was the 13-us epilogue of a 54-us out_proj
        // block (tools/gemm_x3_stamp.py).  Now the pass's residual chunks are requested together, branch-free, before the slab is
        // read, and `m % res_rows` is skipped when the residual has a row of its own for every output row.
        f32x4 rv[NIT];
        if (OUT == 0 && R) {
          auto request = [&](auto nowrap) {
#pragma clang loop unroll(full)
            for (int it = 0; it < NIT; ++it) {
              const int c = it * 64 + lane_e;
              const int row = c / CPRW, ch = c - row * CPRW;
              const int m = min(m0 + wr * TM * 16 + pass * PR + row, p.M - 1);
              int n = n0 + wc * TN * 16 + ch * (16 / ESZ);
              n = n < p.N ? n : 0;
              rv[it] = *(const f32x4*)(R + (long)(decltype(nowrap)::value ? m : m % p.res_rows) * p.ldr + n);
            }
          };
          if (res_nowrap) request(std::true_type{});        // wave-uniform: the integer modulo (~20 instructions) only where it is needed
          else request(std::false_type{});
        }
#pragma clang loop unroll(full)
        for (int it = 0; it < NIT; ++it) {
          const int c = it * 64 + lane_e;
          const int row = c / CPRW, ch = c - row * CPRW;
          const int m = m0 + wr * TM * 16 + pass * PR + row;
          const int n = n0 + wc * TN * 16 + ch * (16 / ESZ);
          f32x4 d = *(const f32x4*)(slab + row * RS + ch * 16);
#ifdef ZH_ABL_SKIP_EPI_STORES  // developer ablation (timing only; round 6): the slab passes stay, the GLOBAL stores of a persistent tile that has a successor
          if (more) { asm volatile("" :: "v"(d)); continue; }        // go — the most that moving the stores to other waves (wave-specialised roles) could hide
#endif
          if (m < p.M && n < p.N) {
            // f32 output: the slab holds fp32, the residual joins before the store
            if (BIAS_LATE) d += bias_late;
            if (OUT == 0 && R) d += rv[it];
            if (OUT == 1) *(f32x4*)((half_t*)p.C + cb + (long)m * p.ldc + n) = d;
            else *(f32x4*)((float*)p.C + cb + (long)m * p.ldc + n) = d;
          }
        }
      }
    }
  } else if (VEC) {
#pragma clang loop unroll(full)
    for (int nt = 0; nt < TN; ++nt) {
      const int n = n0 + (wc * TN + nt) * 16 + fk_e * 4;
      const bool nok = n < p.N;
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (p.bias && nok) bv = *(const f32x4*)(p.bias + n);
#pragma clang loop unroll(full)
      for (int mt = 0; mt < TM; ++mt) {
        const int m = m0 + (wr * TM + mt) * 16 + frow_e;
        if (nok && m < p.M) {
          f32x4 v = acc[nt][mt] * osc + bv;
          if (ACT != ZH_ACT_NONE) {
            v[0] = zh_act(v[0], ACT); v[1] = zh_act(v[1], ACT); v[2] = zh_act(v[2], ACT); v[3] = zh_act(v[3], ACT);
          }
          if (R) v += *(const f32x4*)(R + (long)(m % p.res_rows) * p.ldr + n);
          if (OUT == 1) {
            half4_t h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            *(half4_t*)((half_t*)p.C + cb + (long)m * p.ldc + n) = h;
          } else if (OUT == 0) {
            *(f32x4*)((float*)p.C + cb + (long)m * p.ldc + n) = v;
          } else {
            zh_store_h4((half_t*)p.C + cb + (long)m * p.ldc + n, p.planeC, v);
          }
        }
      }
    }
  } else {   // unaligned / odd-N fallback: scalar stores (rare: odd pixel counts)
#pragma clang loop unroll(full)
    for (int mt = 0; mt < TM; ++mt) {
      const int m = m0 + (wr * TM + mt) * 16 + frow_e;
      const long rrow = R ? (long)(m % p.res_rows) * p.ldr : 0;
#pragma clang loop unroll(full)
      for (int nt = 0; nt < TN; ++nt) {
        const int n = n0 + (wc * TN + nt) * 16 + fk_e * 4;
#pragma clang loop unroll(full)
        for (int e = 0; e < 4; ++e) {
          if (m < p.M && n + e < p.N) {
            float x = acc[nt][mt][e] * osc;
            if (p.bias) x += p.bias[n + e];
            x = zh_act(x, ACT);
            if (R) x += R[rrow + n + e];
            const long ci = cb + (long)m * p.ldc + n + e;
            if (OUT == 1) ((half_t*)p.C)[ci] = (half_t)x;
            else if (OUT == 0) ((float*)p.C)[ci] = x;
            else {
              zh_store_h1((half_t*)p.C + ci, p.planeC, x);
            }
          }
        }
      }
    }
  }
  if (!more) break;
  vb = vb_next;                               // PERS: on to the workgroup's next tile (its operands are already in flight)
  decode(vb, batch, tm, tn);
  m0 = tm * BM; n0 = tn * BN;
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
#ifdef ZH_GEMM_PROBE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ZH_PROBE(3);
#endif
}



int gemm_persist_cus();      // capi.hip: 256, or the developer override (zh_dev_set_gemm_persist)

template <int WM, int WN, int TM, int TN, int STAGES, int OUT, int ACT, int VEC, int SPLIT>
static void launch_one(GemmArgs p, int batch, hipStream_t stream) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  p.nbm = zh_cdiv(p.M, BM);
  p.nbn = zh_cdiv(p.N, BN);
  unsigned nblk = (unsigned)((long)p.nbm * p.nbn * batch);
  p.total = (int)nblk;
  // persistent big plain-fp16 tiles (see PERS in the kernel): one workgroup per CU walks the tiles when there is more than one round of
  // them; never with pos tables (their slice is staged through the ring slot the next tile's prologue would use)
  constexpr bool PERS = !SPLIT && WM * WN == 8 && STAGES == 4 && VEC == 2 && (OUT == 1 || TN == 3);
  if (PERS) {
    const int cus = gemm_persist_cus();                    // the device's CUs; developer override: 0 = off, n = a grid of n workgroups (multiple of 8)
    if (cus > 0 && !p.pos_y && nblk > (unsigned)cus) nblk = (unsigned)cus;
    // (Round 6, measured and removed: staggered starts — eight start phases per XCD, up to 7/8 of a tile apart, for the persistent walks
    //  and, in a second pass, for the first round of every multi-round big-tile launch — so that the CUs' epilogue bursts (fp16 tiles: 128 KB
    //  of stores; fp32 + residual: 256 KB in + 256 KB out, 9.6 TB/s chip-wide if all CUs are in them at once) do not coincide round
    //  after round: config 5 at 6 layers 29.7 - 30.1 ms without, 29.7 - 30.3 with; profiles/r06_gemm_stagger_ab.txt.  What the
    //  epilogue costs is not the coincidence of the bursts.)
  }
  hipLaunchKernelGGL((gemm_f16_kernel<WM, WN, TM, TN, STAGES, OUT, ACT, VEC, SPLIT>), dim3(nblk), dim3(64 * WM * WN), 0, stream, p);
}

// Relative time estimate of a tiling: rounds of the 256-CU chip x time of one round.  With `bpc` blocks resident
// per CU a round takes bpc x the tile's own time; `eff` is the measured relative speed of the tile shape at
// K=768 (256x256: 1.0, 256x192: 0.95, 128x128: 0.8 — tools/gemm_bench.py on MI355X).
// host-side check of the optional pos tables (both or neither)
static inline bool zh_pos_tables_ok(const void* pos_y, const void* pos_x, long ld_pos, int pos_h, int pos_w, int N) {
  if (!pos_y && !pos_x) return true;
  return pos_y && pos_x && pos_h > 0 && pos_w > 0 && ld_pos >= N && ld_pos % 8 == 0 && N % 4 == 0 &&
         (((uintptr_t)pos_y | (uintptr_t)pos_x) & 15) == 0;
}

static inline double tiling_cost(long M, long N, int batch, int BM, int BN, int bpc, double eff) {
  const long tiles = (long)zh_cdiv(M, BM) * zh_cdiv(N, BN) * batch;
  const long slots = 256L * bpc;
  const long rounds = (tiles + slots - 1) / slots;
  return (double)rounds * BM * BN * bpc / eff;
}

// Developer overrides (defined once, in capi.hip): initialised ONCE per process from ZH_GEMM_GROUP_M (super-tile height),
// ZH_GEMM_TILE (forced tile code; unknown codes are an argument error) and ZH_GEMM_TILE_SMALL (applied to M <= 4096 only) —
// never read per launch — and settable at run time through zh_dev_set_gemm_overrides() (tests / tools).
struct GemmDevOverrides { int group_m; int tile; int tile_small; };
const GemmDevOverrides& gemm_dev_overrides();
