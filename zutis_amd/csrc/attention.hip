// Flash attention forward, fp16 operands / fp32 softmax+accumulate, head_dim 64 or 96 (gfx950).
//
// Replaces nn.MultiheadAttention's softmax(QK^T/sqrt(dh))V at networks/clip_arch.py:314-316 (12 heads,
// dh=64, T=1+hw), networks/transformer.py:272-286 (decoder self/cross attention, 8 heads, dh=96) and
// networks/selfmask/vision_transformer.py:110-133 (dh=64, T up to ~6k; the reference materialises
// [B,6,T,T]).  Nothing T x T ever reaches HBM here.
//
// Design: one workgroup = 4 waves = 128 queries of one (image, head); each wave owns 32 queries.
// K/V tiles of 64 keys are register-staged HBM -> LDS (issue loads for tile t+1 before computing tile t).
// "Query on the lane" throughout: S^T = K Q^T via v_mfma_f32_32x32x16_f16 puts query = lane&31 in every
// accumulator register, so the online-softmax max/sum are in-register (one cross-half shuffle) and the
// O^T = V^T P^T accumulator rescale is a per-lane scalar.  K rows are fed to the MFMA in an order that swaps
// key-index bits 2 and 3, which makes the S^T accumulator registers 8s..8s+7 exactly the natural-k fp16
// B-operand of k-step s of the PV product (no lane movement, no LDS round trip for P).  V stays row-major
// [key][d] in LDS and is consumed column-wise through ds_read_b64_tr_b16.
// LDS strides: K rows dh+8 halves (conflict-free ds_read_b128), V rows 96 halves (4 consecutive rows cover
// disjoint 16-dword bank ranges for the transposed read).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

struct AttnArgs {
  const half_t* Q; long ldq, sQ;
  const half_t* K; long ldk, sK;
  const half_t* V; long ldv, sV;
  half_t* O; long ldo, sO;
  int Tq, Tk, H, nqb, groups;
  float scale_log2;
  int causal;       // keys > query masked (CLIP text tower, clip_arch.py:525-531)
  long planeQ, planeK, planeV, planeO;   // X3 kernels: lo planes of the split-pair Q / K / V inputs; planeO != 0: O is written as a split pair
  // key split (zh_attention_f16_splitk): workgroup (group, qb, ks) covers keys [ks * kchunk, (ks + 1) * kchunk) and leaves its
  // UNNORMALISED fp32 accumulators + running max (log2 units) + row sum in the workspace; attn_combine_kernel merges them
  int ksplit, kchunk;
  float* part_o; float* part_m; float* part_l;   // [ksplit][B*Tq][H*dh], [ksplit][B*H][Tq] x 2
#ifdef ZH_ATTN_STAMP
  long long* stamp;   // developer build (tools/attn_stamp.py): [workgroup][wave][16] cycle sums per loop segment + clocks
#endif
};
#ifdef ZH_ATTN_STAMP
static long long* g_attn_stamp = nullptr;
extern "C" void zh_attn_set_stamp(long long* p) { g_attn_stamp = p; }
#define ZH_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const long long n_ = __builtin_amdgcn_s_memtime(); st_acc[i] += n_ - st_prev; st_prev = n_; \
                         __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define ZH_STAMP(i)
#endif

typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) fp16x4* lds_fp16x4_ptr;

#define KT 64
#define VS 96

// VALU diet of the key-tile loop (round 3; every vector instruction costs the SIMD's issue port 4 cycles, v_exp_f32 8, an MFMA 8 of
// its 32 — MI355X_MICROARCH.md "vector-instruction ISSUE cost" — and the softmax of a split-pair tile was ~145 of them beside 24
// MFMAs: issue-bound, not MFMA-bound).  Three pieces: the two key halves of a query meet through v_permlane32_swap instead of
// ds_bpermute (+ 6 address instructions + an LDS round trip per tile); tile loads are SGPR base + 32-bit lane offset with the
// row clamped to the last key (no zero fill, no 64-bit address arithmetic per tile); the running max only moves when it grows
// by more than 2^8 (the accumulator rescale — 32 multiplies per lane — then runs on the first tile and almost never again;
// P <= 256 stays far inside fp16 / the split pair).
// Measured (same box, us; split-pair / fp16 kernels): encoder 79.1 -> 73.3 / 41.8 -> 37.4, cross-attention 123.0 -> 118.0 / 43.7 ->
// 38.2, 518-px encoder 111.8 -> 101.6 / 56.9 -> 50.6, ViT-L/14 1335 -> 1246 / 681 -> 625, SelfMask T = 5505 251 -> 231 / 128 -> 113;
// each piece alone 1 - 3 %.  Not kept: s_setprio 1 around the K.Q^T MFMAs, the P.V MFMAs or the softmax (all within noise),
// -fno-slp-vectorize (the packed multiplies left are the now rare rescale).
// ZH_ATTN_PIPE (developer A/B): -1 = the launcher's rule, 0 / 1 = the split-pair kernels never / always run the pipelined loop
#ifndef ZH_ATTN_PIPE
#define ZH_ATTN_PIPE -1
#endif
#define ZH_ATTN_LAZY_LOG2 8.0f

// max over the two lanes l, l ^ 32 (both get it)
__device__ __forceinline__ float zh_xor32_max(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));   // a: lanes 32.. <- x[0..31]; b: lanes 0..31 <- x[32..]
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float zh_xor32_sum(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}

// X3 = 1 (the reference-equivalent mode): Q, K and V arrive as split pairs (hi = f16(x), lo = f16(x - hi)):
// S = Kh.Qh + Kl.Qh + Kh.Ql in fp32 — the softmax exponent sees fp32-class scores, which is where fp16 operand rounding is
// amplified (|s| * 2^-11 absolute) — and the probabilities are split in registers, O += Vh.Ph + Vl.Ph + Vh.Pl, so short
// key sets (decoder self-attention over 100 queries, small images) are not left with the 2^-11 rounding of single P / V
// values.  32-key tiles (the lo planes of K and V double the LDS tiles and the three-product accumulators the registers):
// three workgroups per CU at dh = 64, two at dh = 96.  The occupancy is checked at build time (build.py MIN_OCCUPANCY reads the
// compiler's resource remarks): asking for 3 through __launch_bounds__ makes hipcc pick a 168-VGPR allocation WITH 16 bytes of
// scratch, while the looser bound compiles to 165 VGPRs and none — so the bound stays loose and the build fails on a regression.
#ifndef ZH_ATTN_ABL
#define ZH_ATTN_ABL 0      // developer ablations (tools/attn_ablate.py): 1 no exp, 2 no P.V, 4 no K.Q^T, 8 no tile traffic and no
#endif                     // barriers, 16 no barriers, 32 barriers only, 64 no LDS stores.  0 in the product build.
// QT = 2 (round 5, developer A/B behind ZH_ATTN_QT=2; split-pair dh = 64 only): a wave owns TWO 32-query tiles.  Every K / V fragment
// read from LDS feeds both tiles' MFMAs and a key tile's loads / stores / barrier serve 256 queries per workgroup: the kernel is
// bound by the SIMD's instruction issue (profiles/NOTES.md round 3), and those are the instructions that do not scale with the scores.
template <int DH, int NWAVE, int X3, int PIPE, int QT = 1>
__global__ __launch_bounds__(64 * NWAVE, X3 ? 2 : (DH == 64 ? 3 : 2)) void attn_f16_kernel(AttnArgs p) {
  static_assert(!PIPE || X3, "the pipelined loop exists for the split-pair kernels");
  static_assert(QT == 1 || (QT == 2 && X3 && !PIPE && DH == 64), "two query tiles per wave: split-pair dh = 64, plain loop");
  constexpr int NT = 64 * NWAVE;
  constexpr int KS = DH + 8;          // K row stride (halves)
  constexpr int NKS = DH / 16;        // k-steps of QK^T
  constexpr int NDT = DH / 32;        // 32-row d tiles of O^T
  constexpr int CPR = DH / 8;         // 16-byte chunks per K/V row
  // keys per tile: 64, except the split-pair kernels — their lo planes double the LDS tiles and their three-product
  // accumulators the registers, so at 64 keys only one workgroup fits a CU (one wave per SIMD, nothing to overlap with);
  // 32-key tiles halve both and two or three workgroups fit
#ifdef ZH_ATTN_X3_KT64
  constexpr int KTT = KT;
#else
  constexpr int KTT = X3 ? 32 : KT;        // (fp16 dh = 96 with 32 keys: cross-attention 44 -> 63 us — 256 workgroups, occupancy is not its limit)
#endif
  constexpr int NU = KTT / 32;        // 32-key slot tiles per key tile
  constexpr int NLD = (KTT * CPR + NT - 1) / NT;              // chunk passes per thread (the last may be partial: dh = 96 x 32 keys)
  constexpr bool LDFULL = (KTT * CPR) % NT == 0;
  __shared__ __attribute__((aligned(16))) half_t sKb[2][KTT * KS];   // double-buffered: one barrier per key tile
  __shared__ __attribute__((aligned(16))) half_t sKl[X3 ? 2 : 1][X3 ? KTT * KS : 8];   // lo plane of K (X3)
  __shared__ __attribute__((aligned(16))) half_t sVb[2][KTT * VS];
  __shared__ __attribute__((aligned(16))) half_t sVl[X3 ? 2 : 1][X3 ? KTT * VS : 8];   // lo plane of V (X3)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef ZH_ATTN_STAMP
  const long long st_e0 = __builtin_amdgcn_s_memtime(), st_re0 = __builtin_amdgcn_s_memrealtime();
#endif
  // XCD-aware block order: workgroups are dealt round-robin to the 8 XCDs (linear id % 8), each with its own L2.  The work items
  // (group = image * heads + head, query block, key split) are numbered group-major and every XCD takes a CONTIGUOUS eighth of
  // that list: the query blocks / key splits of one (image, head) — which stream the same K / V — meet in one L2 (two at a range
  // boundary), and the XCDs' shares differ by at most one workgroup.  (Round 3 dealt whole groups, group % 8 -> XCD: with the 12
  // heads of one image four XCDs got two heads and four got one — the batch-1 encoder ran at the pace of the loaded half.)
  const int id = blockIdx.y * gridDim.x + blockIdx.x;
  const int per_group = p.nqb * p.ksplit;
#ifdef ZH_ATTN_PLAIN_ORDER                               // developer A/B build: query blocks of a group on consecutive ids
  const int item = id;
#else
  const int item = (id & 7) * (int)(gridDim.x >> 3) + (id >> 3);   // the grid is 8 x (items per XCD)
#endif
  if (item >= p.groups * per_group) return;             // surplus workgroups of the last XCD share (whole workgroup, before any barrier)
  const int group = item / per_group;
  const int qbs = item - group * per_group;
  const int qb = qbs / p.ksplit, ks = qbs - qb * p.ksplit;
  const int head = group % p.H, img = group / p.H;
  const int q0 = qb * (32 * NWAVE * QT) + wave * (32 * QT);     // first query of this wave (its tile a starts at q0 + 32 a)
  const int ql = lane & 31, hh = lane >> 5;
  const long hoff = (long)head * DH;

  // this workgroup's keys: all of them, or chunk ks of the key split (kchunk is a multiple of the tile height)
  const int key0 = ks * p.kchunk;
  const int key_end = p.ksplit > 1 ? min(p.Tk, key0 + p.kchunk) : p.Tk;
  const half_t* Q = p.Q + (long)img * p.sQ + hoff;
  const half_t* K = p.K + (long)img * p.sK + hoff;
  const half_t* V = p.V + (long)img * p.sV + hoff;

  // Q fragments (B operand: col = query, k = d)
  half8_t qf[QT][NKS], qfl[QT][X3 ? NKS : 1];
#pragma unroll
  for (int a = 0; a < QT; ++a) {
    int qr = q0 + 32 * a + ql;
    qr = qr < p.Tq ? qr : p.Tq - 1;
    const half_t* qp = Q + (long)qr * p.ldq + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[a][ks] = *(const half8_t*)(qp + 16 * ks);
    if (X3) {
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) qfl[a][ks] = *(const half8_t*)(qp + p.planeQ + 16 * ks);
    }
  }

  f32x16 oacc[QT][NDT];
#pragma unroll
  for (int a = 0; a < QT; ++a)
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[a][d][r] = 0.f;
  float m_run[QT];
#pragma unroll
  for (int a = 0; a < QT; ++a) m_run[a] = -INFINITY;
  // Row sums of P ride the MFMA pipe: one extra "d tile" whose V^T operand is all ones accumulates sum_k P[k][q] in every row
  // of lacc (32 v_add_f32 per key tile leave the VALU, which is the bound; the sum is of the SAME rounded P the numerator uses).
  // Split-pair kernels (X3) keep the row sum on the VALU instead: they are MFMA-bound (three products per score and per P.V
  // element), the ones-operand tile is 4 of their 28 MFMAs per key tile, and P = hi + lo carries 22 bits, so the fp32 sum of the
  // un-rounded exponentials is the same number to 2^-22 — l_run accumulates this lane's 16 keys per tile (v_pk_add_f32), the
  // two key halves (lanes l, l ^ 32) meet once at the end.
  f32x16 lacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) lacc[r] = 0.f;
  float l_run[QT];
#pragma unroll
  for (int a = 0; a < QT; ++a) l_run[a] = 0.f;
  half8_t ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (half_t)1.0f;

  // cooperative tile loads: chunk c -> (row = c / CPR, col chunk = c % CPR).  Tile t+1 is requested before tile t is computed
  // and stored into the other LDS buffer after it (a second register set, i.e. two tiles of load latency cover, measured no
  // gain — the kernel is VALU-bound — and its 16 registers are what keeps dh = 64 at three waves per SIMD).
  struct TileRegs { half8_t k[NLD], v[NLD], kl[X3 ? NLD : 1], vl[X3 ? NLD : 1]; };
  TileRegs ra;
  // per-lane byte offsets of this thread's chunk(s) inside a tile: SGPR base + 32-bit offset; a tile's rows are clamped to the last
  // key of the chunk (a duplicate of a real row: its scores are masked to -inf below, its V rows meet P = 0) — host check:
  // Tk * ld * 2 < 2^32.  K and V are fetched separately: the pipelined loop keeps K one tile ahead of V.
  // (Measured and not kept: a steady-state form with NO vector instruction per tile — constant per-lane offsets, the tile base
  //  advanced in SGPRs — on the theory that the 750-cycle tile-load segment of the stamps was the address instructions queueing
  //  for the vector issue port: encoder 79.8 vs 79.7 us, ViT-L/14 1331 vs 1300, fp16 kernels the same.  The segment is the
  //  loads' own issue into the CU's shared memory pipeline.  Nor did 8-wave workgroups help — 256 queries sharing every tile, one
  //  workgroup per CU with the pipelined loop, a third of the tile bytes per wave: encoder 85.2 vs 75.7 us, ViT-L/14 1438 vs 1255,
  //  518 px 99.1 vs 98.4, only SelfMask's T = 5505 gained (208 vs 219.5).  Dealing the partly filled last query block of every
  //  group after all the full ones (longest first): encoder 77.8 -> 75.7, 518 px 97.3 -> 94.9, but ViT-L/14 1248 -> 1325 (its
  //  light blocks then re-read K / V long after the group's other blocks left the L2).)
  const unsigned ldk2 = (unsigned)p.ldk * 2u, ldv2 = (unsigned)p.ldv * 2u;
  const char* const Kl = (const char*)(K + p.planeK);
  const char* const Vl = (const char*)(V + p.planeV);
  auto load_k = [&](int kbase, TileRegs& r) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = tid + i * NT;
      const int row = c / CPR, cc = c - row * CPR;
      if (LDFULL || c < KTT * CPR) {
        const unsigned ok = (unsigned)min(kbase + row, key_end - 1) * ldk2 + (unsigned)cc * 16u;
        r.k[i] = *(const half8_t*)((const char*)K + ok);
        if (X3) r.kl[i] = *(const half8_t*)(Kl + ok);
      }
    }
  };
  auto load_v = [&](int kbase, TileRegs& r) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = tid + i * NT;
      const int row = c / CPR, cc = c - row * CPR;
      if (LDFULL || c < KTT * CPR) {
        const unsigned ov = (unsigned)min(kbase + row, key_end - 1) * ldv2 + (unsigned)cc * 16u;
        r.v[i] = *(const half8_t*)((const char*)V + ov);
        if (X3) r.vl[i] = *(const half8_t*)(Vl + ov);
      }
    }
  };
  auto store_k = [&](int buf, const TileRegs& r) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = tid + i * NT;
      const int row = c / CPR, cc = c - row * CPR;
      if (LDFULL || c < KTT * CPR) {
        *(half8_t*)(sKb[buf] + row * KS + cc * 8) = r.k[i];
        if (X3) *(half8_t*)(sKl[buf] + row * KS + cc * 8) = r.kl[i];
      }
    }
  };
  auto store_v = [&](int buf, const TileRegs& r) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = tid + i * NT;
      const int row = c / CPR, cc = c - row * CPR;
      if (LDFULL || c < KTT * CPR) {
        *(half8_t*)(sVb[buf] + row * VS + cc * 8) = r.v[i];
        if (X3) *(half8_t*)(sVl[buf] + row * VS + cc * 8) = r.vl[i];
      }
    }
  };

  // K row fed at MFMA row i = lane&31: swap bits 2 and 3 of i
  const int krow = (ql & 0x13) | ((ql & 4) << 1) | ((ql & 8) >> 1);
  // transposed-read lane address pieces: group g = lane>>4, i = lane&15 -> row +(i>>2), col 16*(g&1)+4*(i&3)
  const int tr_row = 8 * hh + ((lane & 15) >> 2);
  const int tr_col = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

  int ntiles = (max(key_end - key0, 0) + KTT - 1) / KTT;
  if (p.causal) {                                           // key tiles entirely above this block's last query are skipped
    const int qlast = min(p.Tq, (qb + 1) * (32 * NWAVE * QT)) - 1;
    ntiles = min(ntiles, qlast / KTT + 1);
  }
  const int qidx = q0 + ql;
  // a wave whose 32 queries all lie beyond Tq (T = 442: two of the last block's four) only helps with the tile loads
  const bool active = q0 < p.Tq;
#ifdef ZH_ATTN_STAMP
  long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
  long long st_prev = st_t0;
#endif

  // ---- S^T(t) = K(t) Q^T from K buffer t & 1 (NU 32-key slot tiles)
  auto qk = [&](int t, f32x16 (&s)[QT][NU]) {
    const half_t* sK = sKb[t & 1];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
#pragma unroll
      for (int a = 0; a < QT; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[a][u][r] = 0.f;
      const half_t* kp = sK + (32 * u + krow) * KS + 8 * hh;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        half8_t kf = *(const half8_t*)(kp + 16 * ks);
        half8_t kl;
        if (X3) kl = *(const half8_t*)(sKl[t & 1] + (32 * u + krow) * KS + 8 * hh + 16 * ks);
#pragma unroll
        for (int a = 0; a < QT; ++a) {                   // (QT = 2: the K fragments just read serve both query tiles)
#if ZH_ATTN_ABL & 4
          s[a][u][ks] += (float)kf[0] * (float)qf[a][ks][0];
#else
          s[a][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[a][ks], s[a][u], 0, 0, 0);
#endif
          if (X3) {
            s[a][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qf[a][ks], s[a][u], 0, 0, 0);
            s[a][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qfl[a][ks], s[a][u], 0, 0, 0);
          }
        }
      }
    }
  };
  // P as packed fp16 pairs: register (r >> 1) & 3 of fragment r >> 3 holds scores r, r + 1 — the MFMA B operand as is
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  struct PFrag { u32x4 hi[NU][2], lo[X3 ? NU : 1][2]; };
  // ---- online softmax of tile t: scores -> P (fp16, or a split pair), accumulators rescaled when the reference point moves
  // (MASKED = false: the caller knows the tile is full and the attention not causal — no branch splits the block, so the
  //  pipelined loop's K.Q^T MFMAs of the next tile can be scheduled in among these instructions)
  auto softmax = [&](int t, f32x16 (&s)[NU], PFrag& P, auto masked, int a = 0) {
    constexpr bool MASKED = decltype(masked)::value;
    const int kbase = key0 + t * KTT;
    // register r of slot tile u holds key kbase + 32u + 16(r>>3) + 8*hh + (r&7)
    float mx = -INFINITY;
    if (MASKED && (kbase + KTT > key_end || (p.causal && kbase + KTT - 1 > q0))) {   // ragged last tile / tiles crossing the diagonal
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kbase + 32 * u + 16 * (r >> 3) + 8 * hh + (r & 7);
          if (key >= key_end || (p.causal && key > qidx)) s[u][r] = -INFINITY;
        }
    }
    {   // 32 scores per lane: four independent v_max3 chains instead of one 32-deep dependent chain
      float m4[4] = {mx, mx, mx, mx};
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int r = 0; r < 16; r += 8) {
          m4[0] = fmaxf(fmaxf(m4[0], s[u][r + 0]), s[u][r + 1]);
          m4[1] = fmaxf(fmaxf(m4[1], s[u][r + 2]), s[u][r + 3]);
          m4[2] = fmaxf(fmaxf(m4[2], s[u][r + 4]), s[u][r + 5]);
          m4[3] = fmaxf(fmaxf(m4[3], s[u][r + 6]), s[u][r + 7]);
        }
      mx = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
    }
    mx = zh_xor32_max(mx);
    ZH_STAMP(2);
    // running max kept in log2 units (scale_log2 > 0 commutes with max): p = 2^(s*c - m) is ONE fma + v_exp_f32.
    // Lazy: the reference point only moves when the tile's max exceeds it by more than 2^8 (any fixed reference gives the same
    // softmax; both key halves of a query see the same mx and m_run, so they decide alike)
    const float m_cand = mx * p.scale_log2;
    const float m_new = m_cand > m_run[a] + ZH_ATTN_LAZY_LOG2 ? m_cand : m_run[a];
    const float alpha = __builtin_amdgcn_exp2f(m_run[a] - m_new);
    m_run[a] = m_new;
    float lsum2[2] = {0.f, 0.f};                             // scalar adds: packed fp32 VALU is an anti-lever beside MFMAs (v_pk_add_f32 ~ +13 issue cycles)
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
#if ZH_ATTN_ABL & 1
        const float e0 = __builtin_fmaf(s[u][r], p.scale_log2, -m_new), e1 = __builtin_fmaf(s[u][r + 1], p.scale_log2, -m_new);
#else
        const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[u][r], p.scale_log2, -m_new));
        const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[u][r + 1], p.scale_log2, -m_new));
#endif
        if (X3) { lsum2[0] += e0; lsum2[1] += e1; }
        const half2_t eh2 = {(half_t)e0, (half_t)e1};      // one v_cvt_pk_f16_f32
        unsigned eh = __builtin_bit_cast(unsigned, eh2);
        if (X3) {
          // lo = f16(e - hi) from the very bits that are used as hi (cf. zh_store_h4, common.h), ONE instruction per value:
          // v_fma_mix{lo,hi}_f16 reads hi as fp16, forms e - hi exactly in fp32 and rounds once (the plain expression costs
          // v_cvt_f32_f16 + v_sub_f32 + half a v_cvt_pk per value: of the softmax's ~26 VALU issue cycles per score, 6 go)
          unsigned el;
          asm("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
              "v_fma_mixhi_f16 %0, -%1, 1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
              : "=&v"(el) : "v"(eh), "v"(e0), "v"(e1));
          P.lo[u][r >> 3][(r >> 1) & 3] = el;
        }
        P.hi[u][r >> 3][(r >> 1) & 3] = eh;
      }
    if (__any(alpha != 1.0f)) {
#pragma unroll
      for (int d = 0; d < NDT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[a][d][r] *= alpha;
      if (!X3) {
#pragma unroll
        for (int r = 0; r < 16; ++r) lacc[r] *= alpha;
      }
    }
    if (X3) l_run[a] = l_run[a] * alpha + (lsum2[0] + lsum2[1]);
    ZH_STAMP(3);
  };
  // ---- O^T += V(t)^T P^T from V buffer t & 1
  auto pv = [&](int t, const PFrag (&P)[QT]) {
    const half_t* sV = sVb[t & 1];
#if ZH_ATTN_ABL & 2
    oacc[0][0][0] += (float)__builtin_bit_cast(half8_t, P[0].hi[0][0])[0] + (float)__builtin_bit_cast(half8_t, P[0].hi[NU - 1][1])[7];
    lacc[0] += 1.0f; l_run[0] += 1.0f;
#else
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (!X3) lacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones, __builtin_bit_cast(half8_t, P[0].hi[u][ks]), lacc, 0, 0, 0);
        const half_t* vp = sV + (32 * u + 16 * ks + tr_row) * VS + tr_col;
#pragma unroll
        for (int d = 0; d < NDT; ++d) {
          fp16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4_ptr)(vp + 32 * d));
          fp16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4_ptr)(vp + 32 * d + 4 * VS));
          half8_t vf;
          __builtin_memcpy(&vf, &lo, 8);
          __builtin_memcpy(((char*)&vf) + 8, &hi, 8);
          half8_t vl;
          if (X3) {
            const half_t* vq = sVl[t & 1] + (32 * u + 16 * ks + tr_row) * VS + tr_col;
            fp16x4 llo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4_ptr)(vq + 32 * d));
            fp16x4 lhi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4_ptr)(vq + 32 * d + 4 * VS));
            __builtin_memcpy(&vl, &llo, 8);
            __builtin_memcpy(((char*)&vl) + 8, &lhi, 8);
          }
#pragma unroll
          for (int a = 0; a < QT; ++a) {                 // (QT = 2: the V fragments just read serve both query tiles)
            const half8_t pf = __builtin_bit_cast(half8_t, P[a].hi[u][ks]);
            oacc[a][d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, oacc[a][d], 0, 0, 0);
            if (X3) {
              oacc[a][d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, pf, oacc[a][d], 0, 0, 0);
              oacc[a][d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, __builtin_bit_cast(half8_t, P[a].lo[u][ks]), oacc[a][d], 0, 0, 0);
            }
          }
        }
      }
#endif
    ZH_STAMP(4);
  };

#if !ZH_ATTN_ABL
  if (PIPE && !p.causal) {
    // ---- software-pipelined loop (split-pair kernels): K runs ONE TILE AHEAD of V, and a wave issues the 12 K.Q^T MFMAs of tile
    // t+1 — which depend on nothing the softmax of tile t touches — in among that softmax's ~95 vector instructions, which
    // then issue in the shadow of the MFMAs instead of next to an idle matrix pipe (in-kernel stamps, tools/attn_stamp.py: with
    // three uncoordinated waves per SIMD the pipe was 62 % busy inside the loop; a wave spent 19 % of a tile in exp / split /
    // pack with no MFMA of its own in flight).  Step t: global loads of K(t+2), V(t+1) -> [S(t+1) || softmax(t)] -> P(t).V(t) ->
    // stores -> barrier.  K(t+2) overwrites the buffer of K(t), last read in step t-1; V(t+1) that of V(t-1), last read in step
    // t-1.  Steps are branch-free inside (full tiles, not causal); the last tile — the only one that can be ragged — is peeled.
    load_k(key0, ra);
    load_v(key0, ra);
    store_k(0, ra);
    store_v(0, ra);
    if (ntiles > 1) { load_k(key0 + KTT, ra); store_k(1, ra); }
    __syncthreads();
    f32x16 sA[QT][NU], sB[QT][NU];
    PFrag P[QT];
    if (active) qk(0, sA);
    __syncthreads();                               // step 0 stores K(2) over K(0): every wave's reads of K(0) come first
    // one step: tile t has a successor; par = t & 1 as a compile-time constant
    auto step = [&](int t, auto par, f32x16 (&cur)[QT][NU], f32x16 (&nxt)[QT][NU]) {
      constexpr int PAR = decltype(par)::value;
      if (t + 2 < ntiles) load_k(key0 + (t + 2) * KTT, ra);
      load_v(key0 + (t + 1) * KTT, ra);
      ZH_STAMP(0);
      if (active) {
        qk(PAR ^ 1, nxt);
        softmax(t, cur[0], P[0], std::false_type{});
        pv(PAR, P);
      }
      if (t + 2 < ntiles) store_k(PAR, ra);
      store_v(PAR ^ 1, ra);
      ZH_STAMP(5);
      __syncthreads();
      ZH_STAMP(6);
    };
    int t = 0;
    for (; t + 2 < ntiles; t += 2) {              // two tiles per trip: buffer indices and the S register sets are compile-time
      step(t, std::integral_constant<int, 0>{}, sA, sB);
      step(t + 1, std::integral_constant<int, 1>{}, sB, sA);
    }
    if (t + 1 < ntiles) {                         // two tiles left
      step(t, std::integral_constant<int, 0>{}, sA, sB);
      if (active) { softmax(t + 1, sB[0], P[0], std::true_type{}); pv(1, P); }
    } else if (active) {                          // one tile left
      softmax(t, sA[0], P[0], std::true_type{});
      pv(0, P);
    }
  } else
#endif
  {
  auto load_tile = [&](int kbase, TileRegs& r) { load_k(kbase, r); load_v(kbase, r); };
  auto store_tile = [&](int buf, const TileRegs& r) { store_k(buf, r); store_v(buf, r); };
  load_tile(key0, ra);
  store_tile(0, ra);
  __syncthreads();
  auto compute = [&](int t) {
    if (active) {
      f32x16 s[QT][NU];
      PFrag P[QT];
      qk(t, s);
      ZH_STAMP(1);
#pragma unroll
      for (int a = 0; a < QT; ++a) softmax(t, s[a], P[a], std::true_type{}, a);   // (a tile beyond Tq computes on a clamped query row and stores nothing)
      pv(t, P);
    }
  };

  // buffer (t+1)&1 was last read in iteration t-1 and every wave passed the barrier that ended it => free to overwrite
#if ZH_ATTN_ABL & 8                                      // ablation: no tile traffic, no barriers
  for (int t = 0; t < ntiles; ++t) compute(t & 1);
#elif ZH_ATTN_ABL & 16                                   // ablation: tile traffic kept, barriers removed (racy: timing only)
  for (int t = 0; t < ntiles; t += 2) {
    if (t + 1 < ntiles) load_tile((t + 1) * KTT, ra);
    compute(t);
    if (t + 1 < ntiles) store_tile(1, ra);
    if (t + 1 >= ntiles) break;
    if (t + 2 < ntiles) load_tile((t + 2) * KTT, ra);
    compute(t + 1);
    if (t + 2 < ntiles) store_tile(0, ra);
  }
#elif ZH_ATTN_ABL & 32                                   // ablation: barriers kept, tile traffic removed
  for (int t = 0; t < ntiles; ++t) { compute(t & 1); __syncthreads(); }
#elif ZH_ATTN_ABL & 64                                   // ablation: global loads kept (consumed by a cheap op), no LDS stores
  for (int t = 0; t < ntiles; ++t) {
    if (t + 1 < ntiles) load_tile((t + 1) * KTT, ra);
    compute(t & 1);
    if (t + 1 < ntiles) m_run[0] += 1e-30f * (float)ra.k[0][0] * (float)ra.v[NLD - 1][7];
    __syncthreads();
  }
#else
  for (int t = 0; t < ntiles; t += 2) {           // two tiles per trip: the LDS buffer index is a compile-time constant
    if (t + 1 < ntiles) load_tile(key0 + (t + 1) * KTT, ra);
    ZH_STAMP(0);
    compute(t);
    if (t + 1 < ntiles) store_tile(1, ra);
    ZH_STAMP(5);
    __syncthreads();
    ZH_STAMP(6);
    if (t + 1 >= ntiles) break;
    if (t + 2 < ntiles) load_tile(key0 + (t + 2) * KTT, ra);
    ZH_STAMP(0);
    compute(t + 1);
    if (t + 2 < ntiles) store_tile(0, ra);
    ZH_STAMP(5);
    __syncthreads();
    ZH_STAMP(6);
  }
#endif
  }
#ifdef ZH_ATTN_STAMP
  if (p.stamp && lane == 0) {
    long long* sp = p.stamp + ((long)id * NWAVE + wave) * 16;
    const long long st_t1 = __builtin_amdgcn_s_memtime(), st_r1 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < 7; ++i) sp[i] = st_acc[i];
    sp[7] = st_t0 - st_e0;                               // prologue: block decode, Q fragments, first tile
    sp[8] = st_t1 - st_t0; sp[9] = st_r1 - st_r0; sp[10] = ntiles; sp[11] = q0 < p.Tq;
    sp[12] = st_re0; sp[13] = st_r1;                     // absolute 100-MHz times: kernel entry, end of the tile loop
  }
#endif

  // f16: every row of lacc holds the full row sum of this lane's query (the MFMA already summed both key halves);
  // split pairs: this lane's half of the keys + the other half's (lane ^ 32)
#pragma unroll
  for (int a = 0; a < QT; ++a) {
  const float l_row = X3 ? zh_xor32_sum(l_run[a]) : lacc[0];
  const int qr = q0 + 32 * a + ql;
  if (p.ksplit > 1) {                                   // partial result of this key chunk: unnormalised O, running max, row sum
    if (qr < p.Tq) {
      float* po = p.part_o + (((long)ks * (p.groups / p.H) + img) * p.Tq + qr) * ((long)p.H * DH) + hoff + 4 * hh;
#pragma unroll
      for (int d = 0; d < NDT; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(f32x4*)(po + 32 * d + 8 * g) = (f32x4){oacc[a][d][4 * g], oacc[a][d][4 * g + 1], oacc[a][d][4 * g + 2], oacc[a][d][4 * g + 3]};
      if (hh == 0) {
        const long mi = ((long)ks * p.groups + group) * p.Tq + qr;
        p.part_m[mi] = m_run[a];
        p.part_l[mi] = l_row;
      }
    }
    continue;
  }
  const float inv = 1.0f / l_row;
  if (qr < p.Tq) {
    half_t* op = p.O + (long)img * p.sO + (long)qr * p.ldo + hoff + 4 * hh;
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 o = {oacc[a][d][4 * g] * inv, oacc[a][d][4 * g + 1] * inv, oacc[a][d][4 * g + 2] * inv, oacc[a][d][4 * g + 3] * inv};
        zh_store_h4(op + 32 * d + 8 * g, p.planeO, o);
      }
  }
  }
#ifdef ZH_ATTN_STAMP
  if (p.stamp && lane == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    p.stamp[((long)id * NWAVE + wave) * 16 + 14] = __builtin_amdgcn_s_memrealtime();   // output stores done
  }
#endif
}

// Merge of the key-split partials: per (row = image * Tq + query, head), m = max_s m_s, w_s = 2^(m_s - m),
// O = sum_s w_s O_s / sum_s w_s l_s — the same arithmetic a single workgroup does when it meets a new running max.
__global__ __launch_bounds__(256) void attn_combine_kernel(const float* part_o, const float* part_m, const float* part_l, half_t* O, long ldo, long sO,
                                                           long planeO, int S, int B, int H, int Tq, int DH) {
  const int D = H * DH, c4 = D / 4;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * Tq * c4) return;
  const int col = (int)(i % c4) * 4;
  const long row = i / c4;
  const int img = (int)(row / Tq), q = (int)(row - (long)img * Tq), head = col / DH;
  float m = -INFINITY;
  for (int s = 0; s < S; ++s) m = fmaxf(m, part_m[(((long)s * B + img) * H + head) * Tq + q]);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float l = 0.f;
  for (int s = 0; s < S; ++s) {
    const long mi = (((long)s * B + img) * H + head) * Tq + q;
    const float w = __builtin_amdgcn_exp2f(part_m[mi] - m);
    l += w * part_l[mi];
    acc += *(const f32x4*)(part_o + (((long)s * B + img) * Tq + q) * D + col) * w;
  }
  const float inv = 1.0f / l;
  zh_store_h4(O + (long)img * sO + (long)q * ldo + col, planeO, acc * inv);
}

static int attention_launch(const void* Q, long ldq, long strideQ, const void* K, long ldk, long strideK,
                            const void* V, long ldv, long strideV, void* O, long ldo, long strideO,
                            int batch, int heads, int Tq, int Tk, int head_dim, float scale, int causal,
                            long planeQ, long planeK, long planeV, long planeO, hipStream_t stream, int ksplit = 1, void* workspace = nullptr,
                            size_t workspace_bytes = 0) {
  ZH_CHECK_ARG(Q && K && V && O, "zh_attention_f16: null operand");
  ZH_CHECK_ARG(batch > 0 && heads > 0 && Tq > 0 && Tk > 0, "zh_attention_f16: bad shape");
  ZH_CHECK_ARG(head_dim == 64 || head_dim == 96, "zh_attention_f16: head_dim %d not in {64, 96}", head_dim);
  ZH_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0 && strideQ % 8 == 0 && strideK % 8 == 0 &&
                   strideV % 8 == 0 && strideO % 4 == 0 && planeQ % 8 == 0 && planeK % 8 == 0 && planeV % 8 == 0 && planeO % 4 == 0,
               "zh_attention_f16: row/batch/plane strides must keep 16-byte (Q,K,V) / 8-byte (O) alignment");
  ZH_CHECK_ARG(((uintptr_t)Q & 15) == 0 && ((uintptr_t)K & 15) == 0 && ((uintptr_t)V & 15) == 0 && ((uintptr_t)O & 7) == 0,
               "zh_attention_f16: misaligned pointer");
  ZH_CHECK_ARG(heads < 65536 && batch < 65536, "zh_attention_f16: heads/batch exceed grid limits");
  // K / V rows are addressed as (per-image, per-head base) + 32-bit byte offset
  ZH_CHECK_ARG((long)Tk * ldk * 2 < (1L << 32) && (long)Tk * ldv * 2 < (1L << 32), "zh_attention_f16: Tk * ldk (ldv) exceeds 2^31 elements");
  ZH_CHECK_ARG((planeQ != 0) == (planeK != 0) && (planeQ != 0) == (planeV != 0),
               "zh_attention_f16: the split-pair mode needs the lo planes of Q, K and V (all three or none)");
  AttnArgs p;
  p.Q = (const half_t*)Q; p.ldq = ldq; p.sQ = strideQ;
  p.K = (const half_t*)K; p.ldk = ldk; p.sK = strideK;
  p.V = (const half_t*)V; p.ldv = ldv; p.sV = strideV;
  p.O = (half_t*)O; p.ldo = ldo; p.sO = strideO;
  p.Tq = Tq; p.Tk = Tk; p.H = heads;
  p.scale_log2 = scale * 1.4426950408889634f;
  p.causal = causal;
  p.planeQ = planeQ; p.planeK = planeK; p.planeV = planeV; p.planeO = planeO;
  // 128-query (4-wave) blocks: each K/V tile is shared four ways.  A 64-query (2-wave) variant was measured slower on
  // every shape of the model (encoder 301 vs 423 TF, cross-attention 230 vs 397 TF) and was dropped.
  const bool x3 = planeQ != 0;
  // developer A/B (round 5): ZH_ATTN_QT=2 gives the split-pair dh = 64 kernel two 32-query tiles per wave (256 queries per workgroup)
  static const int qt_env = [] { const char* e = getenv("ZH_ATTN_QT"); return e ? atoi(e) : 1; }();
  const bool qt2 = qt_env == 2 && x3 && head_dim == 64 && !causal && ksplit <= 1;
  p.nqb = zh_cdiv(Tq, qt2 ? 256 : 128);
  p.groups = heads * batch;
  p.ksplit = 1; p.kchunk = 0; p.part_o = p.part_m = p.part_l = nullptr;
#ifdef ZH_ATTN_STAMP
  p.stamp = g_attn_stamp;
#endif
  if (ksplit > 1) {
    ZH_CHECK_ARG(!causal && ksplit <= 64, "zh_attention_f16_splitk: ksplit %d not in 1..64 (and not for the causal form)", ksplit);
    const int ktt = x3 ? 32 : 64;
    p.ksplit = ksplit;
    p.kchunk = zh_cdiv(zh_cdiv(Tk, ktt), ksplit) * ktt;
    ZH_CHECK_ARG((long)(ksplit - 1) * p.kchunk < Tk, "zh_attention_f16_splitk: ksplit %d leaves an empty key chunk for Tk = %d", ksplit, Tk);
    const size_t no = (size_t)ksplit * batch * Tq * heads * head_dim, nm = (size_t)ksplit * batch * heads * Tq;
    ZH_CHECK_ARG(workspace && workspace_bytes >= (no + 2 * nm) * 4 && ((uintptr_t)workspace & 15) == 0,
                 "zh_attention_f16_splitk: workspace too small or misaligned (%zu < %zu)", workspace_bytes, (no + 2 * nm) * 4);
    p.part_o = (float*)workspace; p.part_m = p.part_o + no; p.part_l = p.part_m + nm;
  }
  const long items = (long)p.groups * p.nqb * p.ksplit;
  ZH_CHECK_ARG(items < (1L << 31) - 8, "zh_attention_f16: grid too large");
  const long nblk = (long)zh_cdiv(items, 8) * 8;        // decoded XCD-aware in the kernel: XCD x takes items [x * nblk / 8, (x + 1) * nblk / 8)
  ZH_CHECK_ARG(nblk < (1L << 31), "zh_attention_f16: grid too large");
  dim3 grid((unsigned)nblk);
  // Split-pair kernels: the software-pipelined loop (K.Q^T of tile t+1 issued in among the softmax of tile t; bit-identical
  // results) needs 193 registers at dh = 64, two waves per SIMD instead of three.  Same box, us, plain -> pipelined: decoder
  // cross-attention (dh = 96, two waves either way) 120.4 -> 103.7; SelfMask T = 5505 (264 workgroups) 240.1 -> 223.2; 518-px
  // encoder (864) 102.3 -> 98.7; 336-px encoder (1536) 76.5 -> 77.4; ViT-L/14 (20480) 1249 -> 1306.  So dh = 96 always takes it,
  // dh = 64 when the grid needs no more rounds of the chip at two workgroups per CU than at three.
  bool pipe = x3 && (head_dim == 96 || zh_cdiv(nblk, 512L) <= zh_cdiv(nblk, 768L));
  if (ZH_ATTN_PIPE >= 0) pipe = x3 && ZH_ATTN_PIPE;
  if (qt2) hipLaunchKernelGGL((attn_f16_kernel<64, 4, 1, 0, 2>), grid, dim3(256), 0, stream, p);
  else if (head_dim == 64) {
    if (x3 && pipe) hipLaunchKernelGGL((attn_f16_kernel<64, 4, 1, 1>), grid, dim3(256), 0, stream, p);
    else if (x3) hipLaunchKernelGGL((attn_f16_kernel<64, 4, 1, 0>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((attn_f16_kernel<64, 4, 0, 0>), grid, dim3(256), 0, stream, p);
  } else {
    if (x3 && pipe) hipLaunchKernelGGL((attn_f16_kernel<96, 4, 1, 1>), grid, dim3(256), 0, stream, p);
    else if (x3) hipLaunchKernelGGL((attn_f16_kernel<96, 4, 1, 0>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((attn_f16_kernel<96, 4, 0, 0>), grid, dim3(256), 0, stream, p);
  }
  ZH_CHECK_LAUNCH("zh_attention_f16");
  if (p.ksplit > 1) {
    const long n4 = (long)batch * Tq * heads * head_dim / 4;
    hipLaunchKernelGGL(attn_combine_kernel, dim3((unsigned)zh_cdiv(n4, 256)), dim3(256), 0, stream, p.part_o, p.part_m, p.part_l, p.O, ldo, strideO,
                       planeO, p.ksplit, batch, heads, Tq, head_dim);
    ZH_CHECK_LAUNCH("zh_attention_f16_splitk (combine)");
  }
  return ZH_OK;
}

// Keys split over `ksplit` workgroups per (image, head, query block) + one merge launch: for few queries against many keys (the
// decoder's cross-attention: 100 queries, 1764 keys -> 256 workgroups, ONE per CU, each streaming 1.35 MB of K / V with a single
// tile of prefetch: bytes in flight bound the kernel at half the HBM rate).  Two workgroups per CU double the bytes in flight.
extern "C" size_t zh_attention_splitk_workspace_size(int batch, int heads, int Tq, int head_dim, int ksplit) {
  return ((size_t)ksplit * batch * Tq * heads * head_dim + 2 * (size_t)ksplit * batch * heads * Tq) * 4;
}
extern "C" int zh_attention_f16_splitk(const void* Q, long ldq, long strideQ, const void* K, long ldk, long strideK,
                                       const void* V, long ldv, long strideV, void* O, long ldo, long strideO,
                                       int batch, int heads, int Tq, int Tk, int head_dim, float scale,
                                       long planeQ, long planeK, long planeV, long planeO, int ksplit, void* workspace, size_t workspace_bytes,
                                       hipStream_t stream) {
  return attention_launch(Q, ldq, strideQ, K, ldk, strideK, V, ldv, strideV, O, ldo, strideO, batch, heads, Tq, Tk, head_dim,
                          scale, 0, planeQ, planeK, planeV, planeO, stream, ksplit, workspace, workspace_bytes);
}

extern "C" int zh_attention_f16(const void* Q, long ldq, long strideQ, const void* K, long ldk, long strideK,
                                const void* V, long ldv, long strideV, void* O, long ldo, long strideO,
                                int batch, int heads, int Tq, int Tk, int head_dim, float scale,
                                long planeQ, long planeK, long planeV, long planeO, hipStream_t stream) {
  return attention_launch(Q, ldq, strideQ, K, ldk, strideK, V, ldv, strideV, O, ldo, strideO, batch, heads, Tq, Tk, head_dim,
                          scale, 0, planeQ, planeK, planeV, planeO, stream);
}

extern "C" int zh_attention_causal_f16(const void* Q, long ldq, long strideQ, const void* K, long ldk, long strideK,
                                       const void* V, long ldv, long strideV, void* O, long ldo, long strideO,
                                       int batch, int heads, int T, int head_dim, float scale,
                                       long planeQ, long planeK, long planeV, long planeO, hipStream_t stream) {
  return attention_launch(Q, ldq, strideQ, K, ldk, strideK, V, ldv, strideV, O, ldo, strideO, batch, heads, T, T, head_dim,
                          scale, 1, planeQ, planeK, planeV, planeO, stream);
}
