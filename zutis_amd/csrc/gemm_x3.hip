// zh_gemm_f16x3: the reference-equivalent GEMM mode.  The reference computes every contraction in fp32
// (networks/clip_arch.py:286-292 keeps LayerNorm in fp32, networks/zutis.py:55 casts the CLIP weights back to fp32);
// gfx950 has no fast fp32 matrix path (v_mfma_f32_*_f32: 157 TFLOP/s dense against 2.5 PFLOP/s for fp16), so fp32-class
// products are built from fp16 MFMAs: each operand x is carried as the pair hi = f16(x), lo = f16(x - hi) — 22
// significand bits — and  A.W = Ah.Wh + Ah.Wl + Al.Wh  accumulates in fp32 (dropped Al.Wl term: 2^-22 relative).
// Weights are packed as W * 2^s (s chosen per matrix so that lo stays a normal fp16 number); `out_scale` = 2^-s is applied
// to the accumulator before the bias.  Kernel: gemm_kernel.h with SPLIT = 1.
#include "gemm_kernel.h"
#include "gemm_skinny.h"
#ifndef RING64
#define RING64 6      // ring depth of the 64 x 64 tile (developer A/B, round 3: -DRING64=8 — config 3 forward 3.02 ms either way, batch-1 336 px 2.50 vs 2.55 ms: not bound by the slices in flight)
#endif
#ifndef ZH_SKINNY_MAX_ROWS
#define ZH_SKINNY_MAX_ROWS 128
#endif
#ifndef ZH_SKINNY_MAX_BLOCKS
#define ZH_SKINNY_MAX_BLOCKS 512
#endif
#ifdef ZH_GEMM_PROBE
extern "C" void zh_gemm_x3_set_probe(long long* p) { g_probe = p; }   // developer build (tools/gemm_x3_stamp.py)
#endif

template <int WM, int WN, int TM, int TN, int STAGES, int VEC, int SP = 1>
static bool launch_x3(const GemmArgs& p, int batch, int out_kind, hipStream_t stream) {
  const int key = out_kind * 8 + p.act;
  switch (key) {
    case 0 + ZH_ACT_NONE: launch_one<WM, WN, TM, TN, STAGES, 0, ZH_ACT_NONE, VEC, SP>(p, batch, stream); return true;
    case 0 + ZH_ACT_SIGMOID: launch_one<WM, WN, TM, TN, STAGES, 0, ZH_ACT_SIGMOID, VEC, SP>(p, batch, stream); return true;
    case 8 + ZH_ACT_NONE: launch_one<WM, WN, TM, TN, STAGES, 1, ZH_ACT_NONE, VEC, SP>(p, batch, stream); return true;
    case 16 + ZH_ACT_NONE: launch_one<WM, WN, TM, TN, STAGES, 2, ZH_ACT_NONE, VEC, SP>(p, batch, stream); return true;
    default: break;
  }
  if (VEC == 2) {   // activations feeding another GEMM only occur on 16-byte aligned rows
    switch (key) {
      case 8 + ZH_ACT_QUICKGELU: launch_one<WM, WN, TM, TN, STAGES, 1, ZH_ACT_QUICKGELU, 2, SP>(p, batch, stream); return true;
      case 8 + ZH_ACT_RELU: launch_one<WM, WN, TM, TN, STAGES, 1, ZH_ACT_RELU, 2, SP>(p, batch, stream); return true;
      case 8 + ZH_ACT_GELU_ERF: launch_one<WM, WN, TM, TN, STAGES, 1, ZH_ACT_GELU_ERF, 2, SP>(p, batch, stream); return true;
      case 16 + ZH_ACT_QUICKGELU: launch_one<WM, WN, TM, TN, STAGES, 2, ZH_ACT_QUICKGELU, 2, SP>(p, batch, stream); return true;
      case 16 + ZH_ACT_RELU: launch_one<WM, WN, TM, TN, STAGES, 2, ZH_ACT_RELU, 2, SP>(p, batch, stream); return true;
      case 16 + ZH_ACT_GELU_ERF: launch_one<WM, WN, TM, TN, STAGES, 2, ZH_ACT_GELU_ERF, 2, SP>(p, batch, stream); return true;
      default: break;
    }
  }
  return false;
}

extern "C" int zh_gemm_f16x3(const void* A, long lda, long strideA, long planeA, const void* W, long ldw, long strideW, long planeW,
                             void* C, long ldc, long strideC, long planeC, int out_kind, float out_scale,
                             const float* bias, const float* residual, long ldr, long strideR, int res_rows,
                             const void* pos_y, const void* pos_x, long ld_pos, int pos_h, int pos_w, int pos_f16,
                             int act, int M, int N, int K, int batch, int flags, hipStream_t stream) {
  ZH_CHECK_ARG(A && W && C, "zh_gemm_f16x3: null operand");
  ZH_CHECK_ARG(M > 0 && N > 0 && K > 0 && batch > 0, "zh_gemm_f16x3: bad shape M=%d N=%d K=%d batch=%d", M, N, K, batch);
  ZH_CHECK_ARG(K % 64 == 0, "zh_gemm_f16x3: K=%d must be a multiple of 64", K);
  ZH_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && strideA % 8 == 0 && strideW % 8 == 0 && planeA % 8 == 0 && planeW % 8 == 0,
               "zh_gemm_f16x3: lda/ldw/strides/planes must be multiples of 8 halves (16-byte rows)");
  // planeW == 0: W has no lo plane — every packed value is an fp16 number (the released CLIP weights after the reference's
  // convert_weights) — and the kernel skips the zero product (SPLIT = 2 in gemm_kernel.h: bit-identical results, 2/3 of the MFMAs)
  ZH_CHECK_ARG(planeA != 0, "zh_gemm_f16x3: A must be a split pair (plane offset of the lo half)");
  const bool x2 = planeW == 0;
  ZH_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0, "zh_gemm_f16x3: A/W must be 16-byte aligned");
  ZH_CHECK_ARG(act >= 0 && act <= 4, "zh_gemm_f16x3: bad activation %d", act);
  // the kernel addresses an operand row as base + (32-bit element offset): keep one batch item's A / W below 2^32 elements
  ZH_CHECK_ARG((long)(M - 1) * lda + K <= 0xFFFFFFFFL && (long)(N - 1) * ldw + K <= 0xFFFFFFFFL,
               "zh_gemm_f16x3: an operand exceeds 2^32 elements per batch item (M=%d lda=%ld N=%d ldw=%ld)", M, lda, N, ldw);
  ZH_CHECK_ARG(out_kind >= 0 && out_kind <= 2, "zh_gemm_f16x3: out_kind %d not in {0 f32, 1 f16, 2 split pair}", out_kind);
  ZH_CHECK_ARG(out_kind != 2 || (planeC != 0 && planeC % 4 == 0), "zh_gemm_f16x3: split output needs planeC (multiple of 4)");
  // residual[m % res_rows] is added in fp32 after the activation; for fp16 / split-pair outputs (a row-periodic additive table:
  // the decoder's query_pos terms) that is BEFORE the one rounding, and only without an activation
  ZH_CHECK_ARG(!residual || (res_rows > 0 && (out_kind == 0 || act == ZH_ACT_NONE)),
               "zh_gemm_f16x3: residual needs res_rows > 0 and, for fp16 / split outputs, no activation");
  ZH_CHECK_ARG(zh_pos_tables_ok(pos_y, pos_x, ld_pos, pos_h, pos_w, N), "zh_gemm_f16x3: pos tables need both pointers 16-byte aligned, "
               "pos_h, pos_w > 0, ld_pos %% 8 == 0 and N %% 4 == 0");
  ZH_CHECK_ARG(out_scale > 0.0f, "zh_gemm_f16x3: out_scale must be positive");
  GemmArgs p;
  p.A = (const half_t*)A; p.lda = lda; p.sA = strideA; p.planeA = planeA;
  p.W = (const half_t*)W; p.ldw = ldw; p.sW = strideW; p.planeW = planeW;
  p.C = C; p.ldc = ldc; p.sC = strideC; p.planeC = planeC; p.out_scale = out_scale;
  p.bias = bias; p.R = residual; p.ldr = ldr; p.sR = strideR; p.res_rows = res_rows;
  p.pos_y = pos_y; p.pos_x = pos_x; p.ld_pos = ld_pos; p.pos_hw = pos_h * pos_w; p.pos_w = pos_w; p.pos_f16 = pos_f16;
  p.M = M; p.N = N; p.K = K; p.act = act; p.nbm = p.nbn = 0;
  p.group_m = gemm_dev_overrides().group_m;
#ifdef ZH_GEMM_PROBE
  p.probe = g_probe;
#endif
  const int esz = out_kind == 0 ? 4 : 2;
  p.vec_ok = (N % 4 == 0) && (ldc % 4 == 0) && (strideC % 4 == 0) && (((uintptr_t)C & (4 * esz - 1)) == 0) &&
             (!bias || ((uintptr_t)bias & 15) == 0) &&
             (!residual || (ldr % 4 == 0 && strideR % 4 == 0 && ((uintptr_t)residual & 15) == 0));
  ZH_CHECK_ARG((long)zh_cdiv(M, 64) * zh_cdiv(N, 64) * batch < (1L << 31), "zh_gemm_f16x3: grid too large");
  // the LDS-staged epilogue reads the slab in 16-byte chunks: f32 / split rows need N % 4 == 0 (vec_ok), f16 rows N % 8 == 0
  const bool wide_ok = p.vec_ok && (((uintptr_t)C & 15) == 0) && ((ldc * esz) % 16 == 0) && ((strideC * esz) % 16 == 0) &&
                       (out_kind == 1 ? (N % 8 == 0 && !residual) : true) &&      // fp16 slab: a residual takes the direct-store path
                       (out_kind != 2 || (ldc % 8 == 0 && planeC % 8 == 0 && N % 8 == 0));
  // 256 x 128 or 192 x 128 (8 waves, 144 / 120 KiB ring, one block per CU; the 192-row tile quantises N = 768 GEMMs into
  // 1.73 rounds of the chip instead of 1.31) or 128 x 64 (4 waves, two blocks per CU) for small problems
  // ... or, for M >= 4096 rows, the two-slot big tiles (8 waves, 2 x 4): 256 x 256 (waves of 128 x 64, 64 KiB per K slice) and
  // 192 x 256 (waves of 96 x 64, 56 KiB) — a third fewer staged bytes per MFMA than 256 x 128, which is what bounds the 3-slot
  // loop (tools/gemm_x3_probe.sh).  256 x 256 serves the wide GEMMs (QKV, c_fc, the K / V projections); 192 x 256 turns the
  // N = 768 ones (out_proj, c_proj) into ONE round of 222 tiles where 192 x 128 needs two rounds of 444.
  // (Measured and not kept, round 3: a 128 x 128 two-slot tile at TWO blocks per CU, so that one block's epilogue overlaps the
  //  other's K loop — K / V 514 us against 454 for 256 x 256: the extra staged bytes per MFMA cost more than the overlap gives;
  //  and starting the first wave of tiles out of phase — no change: the K = 256 projections write their 1.04 GB at the rate a
  //  pure store kernel reaches for any store shape, 5.4 - 5.8 TB/s (tools/micro/store_pattern.hip), only not while MFMAs run.
  //  Also timed and not kept: weights read as if packed slice-major [K/32][N][32] (every W piece one contiguous KiB instead of 16
  //  half lines: c_proj 175 -> 168 us, QKV / c_fc / K / V within the noise), `s_setprio` around the MFMA sweeps (-20 %) and a
  //  static priority difference between the two waves of a SIMD (no change).  A 224 x 256 tile for c_fc — 768 tiles = three FULL
  //  rounds of the chip instead of 2.6 rounds of 256 x 256: 184.3 against 183 - 190 us, no change (and hipcc spills 4 registers
  //  at TM = 7): under the power limit a round that leaves CUs idle lets the others clock higher.  Touching the tile's residual
  //  lines eight slices before the end of the K loop, so that the fp32-residual epilogue finds them in the L2: out_proj 58.1 ->
  //  63 - 65 us, c_proj 175 -> 179 — the loads stall the slice that retires them and the residual was cache-resident anyway.)
  // few-row GEMMs (the decoder at batch 1, ffn2, SelfMask's 20 queries): gemm_skinny.h — operands straight into fragments, K split
  // over the four waves of a block, no ring.  Tile code 32 forces it for any M (tests), any other forced code disables it.
  {
    const int forced_tile = gemm_dev_overrides().tile;
    // ZH_GEMM_FIXED_K_ORDER: the caller compares results of calls with different M / N bit for bit (sharded retrieval): the LDS-ring
    // family only, whose K order is the same for every tile.  Otherwise: few rows AND a grid small enough that the kernel's re-reads of
    // the row block (once per 32 output columns) do not matter — the class-logit GEMM of a 32-image batch (81 x 1764 x 512 x 32 =
    // 5376 workgroups) took 113 us here against ~20 on the ring.
    const int max_rows = forced_tile == 32 ? (1 << 30) : ((forced_tile || (flags & 1)) ? 0 : ZH_SKINNY_MAX_ROWS);
    const long sk_blocks = (long)zh_cdiv(N, 32) * zh_cdiv(M, 32) * batch;
    if (gemm_skinny_ok(p, batch, p.vec_ok, max_rows) && (forced_tile == 32 || sk_blocks <= ZH_SKINNY_MAX_BLOCKS)) {
      if (x2) launch_skinny<1>(p, batch, out_kind, stream);
      else launch_skinny<2>(p, batch, out_kind, stream);
      ZH_CHECK_LAUNCH("zh_gemm_f16x3");
      return ZH_OK;
    }
  }
  const double c256 = tiling_cost(M, N, batch, 256, 128, 1, 1.0);
  const double c192 = tiling_cost(M, N, batch, 192, 128, 1, 0.95);
  const double c64 = tiling_cost(M, N, batch, 128, 64, 2, 0.7);
  int pick = (c64 < c256 && c64 < c192) ? 64 : (c192 < c256 ? 192 : 256);
  if (M >= 4096) {
    const double cbig = tiling_cost(M, N, batch, 256, 256, 1, 1.25);
    const double c448 = tiling_cost(M, N, batch, 192, 256, 1, 1.2);
    const double cur = pick == 64 ? c64 : (pick == 192 ? c192 : c256);
    if (cbig < cur && cbig <= c448) pick = 512;
    else if (c448 < cur) pick = 448;
  }
  // few-row GEMMs whose 128 x 64 tiling is a bit more than one tile per CU (the decoder's 3200 x 768: 300 tiles — the CUs with two
  // tiles stream 1.2 MB of operands at the ~61 GB/s a CU's LDS-DMA sustains = 19 us, and that is the kernel): 128 x 96 tiles
  // (3-slot ring, one block per CU) make it 200 tiles of 0.69 MB, ONE per CU.  Eight waves of 32 x 48 (two per SIMD) rather than
  // four of 64 x 48: the fp32-residual epilogue has twice the waves to hide its loads behind (24.5 -> 21.8 us; split out 20.2 both)
  // (a 4- or 5-slot ring for this tile — three or four slices in flight instead of two: 19.8 / 19.8 / 20.5 us, no change)
  if (pick == 64 && M <= 4096 && N % 96 == 0) {
    const long t64 = (long)zh_cdiv(M, 128) * zh_cdiv(N, 64) * batch, t96 = (long)zh_cdiv(M, 128) * (N / 96) * batch;
    if (t64 > 256 && t96 <= 256) pick = 96;
  }
  // few-tile GEMMs (batch-1 inference, the COCO-20K evaluation's regime: M = 442 tokens -> 48 .. 192 tiles of 128 x 64 on 256 CUs):
  // they are bound by the load latency of a 3-slot ring, not by arithmetic — 64 x 64 tiles on a 6-deep ring put more CUs to
  // work and keep five slices in flight (the fp16 kernel's `3064` tile, for the split-pair operands)
  // (only while the 64 x 64 tiles still fit ONE round of the chip, one block per CU: c_fc at M = 442 is 336 such tiles and ran
  //  18.3 -> 23.6 us with them)
  if (pick == 64 && (long)zh_cdiv(M, 64) * zh_cdiv(N, 64) * batch <= 256) pick = 3064;
  // Round 5: these one-round tiles move their operands at (bytes in flight per CU) / (load latency) — 64-k slices (128-B row pieces: 85
  // instead of 52 - 57 GB/s of request rate) on a ring deep enough to keep TWO or more of them in flight: the 64 x 64 tile on four slots
  // of 32 KiB (was six 32-k slots: out_proj at one image 12.3 -> 11.1 us, QKV / c_fc at 442 tokens 13.6 -> 12.2 / 24.2 -> 21.2), and a
  // 128 x 64 tile on three slots of 48 KiB where ITS tiles are one round and the 64 x 64 ones are not (1201 x 1536 x 768: 18.8 -> 15.1 us).
  // The 128 x 96 / 128 x 128 tiles of the one-image QKV / c_fc stage 56 / 64 KiB per 64-k slice: two slots, one slice in flight — a tie
  // with their 32-k forms (6496); fetching their A lo fragments straight into registers to make room (40 / 48 KiB slots) was slower
  // (QKV 24.4 -> 26.4 us: the duplicate requests of the waves that share rows cost the request rate more than the deeper ring returns).
#ifndef ZH_X3_ROUND4_SMALL_TILES   // developer A/B (tools/build_variant_lib.sh): the one-round tiles as selected until round 5
  if (pick == 64 && (long)zh_cdiv(M, 128) * zh_cdiv(N, 64) * batch <= 256) pick = 6464;
#else
  if (pick == 3064) pick = 3066;
#endif
  const int forced = gemm_dev_overrides().tile;
  if (forced == 64 || forced == 96 || forced == 192 || forced == 256 || forced == 512 || forced == 448 || forced == 3064) pick = forced;
  // M ~ 1200 rows (one image at native resolution: c_fc 1201 x 3072, the split-K planes of c_proj / out_proj): 128 x 64 tiles are 480
  // workgroups, two per CU, each staging its own A panel; 128 x 128 tiles (8 waves of 32 x 64, 4 slots) are 240, ONE round with a third
  // fewer staged bytes: c_fc 29.4 -> 24.6 us, c_proj planes (S = 4) 28.8 -> 24.8, out_proj planes 12.2 -> 11.0 (round 4, same box;
  // 128 x 96 on 5 slots, 96 x 128, 160 x 128 and 4-wave 128 x 128 were timed with it and dropped: within 3 % or slower)
  if (pick == 64 && (long)zh_cdiv(M, 128) * zh_cdiv(N, 128) * batch <= 256 && (long)zh_cdiv(M, 128) * zh_cdiv(N, 64) * batch > 256) pick = 1288;
  // (developer A/B, code 6496: the 128 x 96 tile on 64-k slices — 128-B row pieces, two slots, gemm_kernel.h K64.  A CU's pure LDS-DMA
  //  stream runs 85 GB/s in 128-B pieces against 52 - 57 in 64-B pieces (tools/micro/dma_stream.hip), but inside the GEMM the form with
  //  ONE slice in flight ties with the 3-slot 32-k form: QKV at one image 21.9 - 22.8 us against 22.4 - 23.5 over four runs, K loop
  //  18.2 against 13.8 us on a cold weight (profiles/NOTES.md round 4): measured, tested, not selected.)
  if (forced == 1288 || forced == 6496 || forced == 6464 || forced == 3066 || ((forced == 7096 || forced == 7128) && !x2 && !pos_y)) pick = forced;
  // The one-image tiles above 48 KiB per 64-k slice (QKV's 128 x 96: 56 KiB, c_fc's and the split-K planes' 128 x 128: 64 KiB) on a CIRCULAR ring
  // of 160 one-KiB pieces (gemm_kernel.h FRAC): behind every barrier one slice's worth of pieces — the tail of slice kt + 1, then the head
  // of slice kt + 2 — refills what slice kt - 1 left, 48 .. 104 KiB ahead of the reads instead of the 56 of two slots.  QKV at one image
  // 23.5 -> 21.8 us, the decoder's 3200 x 768 x 768 21.1 -> 20.8 (r05_gemm_k64_deep.txt; 128 x 128: c_fc 28.9 -> 27.5, c_proj planes 27.6 -> 26.5):
  // the pieces that go out ONE step ahead (an eighth of a 128 x 96 slice, half of a 128 x 128 one) still wait out their latency.
#ifndef ZH_X3_ROUND4_SMALL_TILES
  if (!forced && !x2 && !pos_y) {
    if (pick == 96) pick = 7096;
    // (7128, the 128 x 128 tile on the same circle: half of every slice still goes out one step ahead — c_fc 28.9 -> 27.5 us back to
    //  back, but 26.5 -> 26.4 per launch inside one image's dependent chain (r05_c3_launch_list.txt): developer code, not selected)
  }
#endif
  if ((forced == 5122 || forced == 5124 || forced == 4484) && x2) pick = forced;   // developer A/B: the x2 256 x 256 tile on TWO slots (2 x 4 waves of 128 x 64) / on three as 2 x 4 waves of 128 x 64
  // the two-slot tiles address operand rows as SGPR base + 32-bit per-lane BYTE offset
  if ((pick == 512 || pick == 448 || pick == 5122 || pick == 5124 || pick == 4484) && ((long)(M - 1) * lda + K > 0x7FFFFFFFL || (long)(N - 1) * ldw + K > 0x7FFFFFFFL)) pick = 256;
  // Tail peel.  A big-tile GEMM whose tile count is a few tiles more than whole rounds of the 256-CU chip pays a full round for
  // them (config 5's out_proj / c_proj, 147712 x 1024: 2308 tiles of 256 x 256 = 9 rounds + 4 tiles, i.e. 10 rounds: +10 %).  When
  // the surplus is at most a quarter round, the last m-tile rows that hold it are peeled into a second call on the row range
  // [M', M) — its own, small-tile choice; same stream, same arithmetic per output element (the K order of a tile does not depend
  // on the tile shape: results are bitwise those of one launch) — and the main launch is left with whole rounds.
  // Only where row offsets are plain pointer offsets: no batch, no pos tables, a residual with a row of its own per output row.
  if ((pick == 512 || pick == 448) && wide_ok && batch == 1 && !pos_y && (!residual || res_rows >= M) && !forced && gemm_dev_overrides().tile_small == 0) {
    const int BMp = pick == 512 ? 256 : 192;
    const long nbm = zh_cdiv(M, BMp), nbn = zh_cdiv(N, 256), tiles = nbm * nbn, rem = tiles % 256;
    if (tiles > 4 * 256 && rem > 0 && rem <= 64) {
      const long r = (rem + nbn - 1) / nbn;                       // m-tile rows to peel
      const long M1 = (nbm - r) * BMp;                             // rows that stay: whole tiles, whole rounds (or just under)
      if (r * nbn <= 64 && M1 > 0 && M1 < M) {
        const long esz = out_kind == 0 ? 4 : 2;
        int rc = zh_gemm_f16x3(A, lda, strideA, planeA, W, ldw, strideW, planeW, C, ldc, strideC, planeC, out_kind, out_scale, bias,
                               residual, ldr, strideR, residual ? res_rows : 0, pos_y, pos_x, ld_pos, pos_h, pos_w, pos_f16, act,
                               (int)M1, N, K, 1, flags, stream);
        if (rc != ZH_OK) return rc;
        return zh_gemm_f16x3((const char*)A + M1 * lda * 2, lda, strideA, planeA, W, ldw, strideW, planeW,
                             (char*)C + M1 * ldc * esz, ldc, strideC, planeC, out_kind, out_scale, bias,
                             residual ? residual + M1 * ldr : nullptr, ldr, strideR, residual ? (int)(res_rows - M1) : 0,
                             pos_y, pos_x, ld_pos, pos_h, pos_w, pos_f16, act, (int)(M - M1), N, K, 1, flags, stream);
      }
    }
  }
  bool ok;
  if (x2) {
    // the big tiles stage 48 KiB per slice without the W lo rows: three slots (two slices of prefetch) fit the LDS
    if (!p.vec_ok) ok = launch_x3<2, 2, 4, 2, 3, 0, 2>(p, batch, out_kind, stream);
    else if (!wide_ok) ok = launch_x3<2, 2, 4, 2, 3, 1, 2>(p, batch, out_kind, stream);
    // 256 x 256 as 4 x 2 waves of 64 x 128: an A fragment costs two LDS reads (hi, lo), a W fragment one — 16 reads per slice and wave
    // instead of the 20 of 128 x 64 waves (5124): L/14 qkv 1604 -> 1568 us, c_proj 2266 -> 2099, c_fc 2219 -> 2204 (same box)
    else if (pick == 512) ok = launch_x3<4, 2, 4, 8, 3, 2, 2>(p, batch, out_kind, stream);
    else if (pick == 5122) ok = launch_x3<2, 4, 8, 4, 2, 2, 2>(p, batch, out_kind, stream);
    else if (pick == 5124) ok = launch_x3<2, 4, 8, 4, 3, 2, 2>(p, batch, out_kind, stream);
    else if (pick == 4484) ok = launch_x3<4, 2, 3, 8, 3, 2, 2>(p, batch, out_kind, stream);   // 192 x 256 as 4 x 2 waves of 48 x 128
    else if (pick == 448) ok = launch_x3<2, 4, 6, 4, 3, 2, 2>(p, batch, out_kind, stream);
    else if (pick == 256) ok = launch_x3<4, 2, 4, 4, 3, 2, 2>(p, batch, out_kind, stream);
    else if (pick == 192) ok = launch_x3<4, 2, 3, 4, 4, 2, 2>(p, batch, out_kind, stream);   // 4 slots: the epilogue slabs need 102 KiB
    else if (pick == 96) ok = launch_x3<4, 2, 2, 3, 3, 2, 2>(p, batch, out_kind, stream);
    else if (pick == 3064) ok = launch_x3<2, 2, 2, 2, 4, 2, 2>(p, batch, out_kind, stream);
    else if (pick == 3066) ok = launch_x3<2, 2, 2, 2, RING64, 2, 2>(p, batch, out_kind, stream);
    else if (pick == 6464) ok = launch_x3<4, 2, 2, 2, 3, 2, 2>(p, batch, out_kind, stream);
    else if (pick == 1288) ok = launch_x3<4, 2, 2, 4, 4, 2, 2>(p, batch, out_kind, stream);
    else if (pick == 6496) ok = launch_x3<4, 2, 2, 3, 2, 2, 2>(p, batch, out_kind, stream);
    else ok = launch_x3<2, 2, 4, 2, 3, 2, 2>(p, batch, out_kind, stream);
  } else
  if (!p.vec_ok) ok = launch_x3<2, 2, 4, 2, 3, 0>(p, batch, out_kind, stream);
  else if (!wide_ok) ok = launch_x3<2, 2, 4, 2, 3, 1>(p, batch, out_kind, stream);
  else if (pick == 512) ok = launch_x3<2, 4, 8, 4, 2, 2>(p, batch, out_kind, stream);
  else if (pick == 448) ok = launch_x3<2, 4, 6, 4, 2, 2>(p, batch, out_kind, stream);
  else if (pick == 256) ok = launch_x3<4, 2, 4, 4, 3, 2>(p, batch, out_kind, stream);
  else if (pick == 192) ok = launch_x3<4, 2, 3, 4, 3, 2>(p, batch, out_kind, stream);
  else if (pick == 96) ok = launch_x3<4, 2, 2, 3, 3, 2>(p, batch, out_kind, stream);   // 8 waves of 32 x 48
  else if (pick == 3064) ok = launch_x3<2, 2, 2, 2, 4, 2>(p, batch, out_kind, stream);   // 64 x 64 on 64-k slices, four slots (three slices in flight)
  else if (pick == 3066) ok = launch_x3<2, 2, 2, 2, RING64, 2>(p, batch, out_kind, stream);   // developer A/B: 64 x 64 on 32-k slices, six slots (the form until round 5)
  else if (pick == 1288) ok = launch_x3<4, 2, 2, 4, 4, 2>(p, batch, out_kind, stream);   // 128 x 128, 8 waves of 32 x 64, 4 slots
  else if (pick == 6496) ok = launch_x3<4, 2, 2, 3, 2, 2>(p, batch, out_kind, stream);   // K64 (64-k slices, two slots): 128 x 96, 8 waves of 32 x 48
  else if (pick == 6464) ok = launch_x3<4, 2, 2, 2, 3, 2>(p, batch, out_kind, stream);   // K64 on THREE slots: 128 x 64, 8 waves of 32 x 32
  else if (pick == 7096) ok = launch_x3<4, 2, 2, 3, 4, 2>(p, batch, out_kind, stream);   // K64 on a circular ring of 160 pieces: 128 x 96
  else if (pick == 7128) ok = launch_x3<4, 2, 2, 4, 3, 2>(p, batch, out_kind, stream);   // ... 128 x 128
  else ok = launch_x3<2, 2, 4, 2, 3, 2>(p, batch, out_kind, stream);
  ZH_CHECK_ARG(ok, "zh_gemm_f16x3: (out_kind=%d, act=%d) is not an instantiated epilogue", out_kind, act);
  ZH_CHECK_LAUNCH("zh_gemm_f16x3");
  return ZH_OK;
}
