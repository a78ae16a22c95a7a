// zh_gemm_f16: fp16-operand / fp32-accumulate instantiations of the MFMA GEMM (kernel + design notes: gemm_kernel.h).
#define ZH_GEMM_MAIN
#include "gemm_kernel.h"

// Instantiated (out type, activation) pairs = the ones the hot path uses; anything else is an argument error.
template <int WM, int WN, int TM, int TN, int STAGES, int VEC>
static bool launch_gemm(const GemmArgs& p, int batch, int out_f16, hipStream_t stream) {
  const int key = out_f16 * 8 + p.act;
  switch (key) {
    case 0 + ZH_ACT_NONE: launch_one<WM, WN, TM, TN, STAGES, 0, ZH_ACT_NONE, VEC, 0>(p, batch, stream); return true;
    case 0 + ZH_ACT_SIGMOID: launch_one<WM, WN, TM, TN, STAGES, 0, ZH_ACT_SIGMOID, VEC, 0>(p, batch, stream); return true;
    case 8 + ZH_ACT_NONE: launch_one<WM, WN, TM, TN, STAGES, 1, ZH_ACT_NONE, VEC, 0>(p, batch, stream); return true;
    case 8 + ZH_ACT_QUICKGELU: launch_one<WM, WN, TM, TN, STAGES, 1, ZH_ACT_QUICKGELU, VEC, 0>(p, batch, stream); return true;
    case 8 + ZH_ACT_RELU: launch_one<WM, WN, TM, TN, STAGES, 1, ZH_ACT_RELU, VEC, 0>(p, batch, stream); return true;
    case 8 + ZH_ACT_GELU_ERF: launch_one<WM, WN, TM, TN, STAGES, 1, ZH_ACT_GELU_ERF, VEC, 0>(p, batch, stream); return true;
    default: return false;
  }
}

extern "C" int zh_gemm_f16(const void* A, long lda, long strideA, const void* W, long ldw, long strideW,
                           void* C, long ldc, long strideC, int out_f16,
                           const float* bias, const float* residual, long ldr, long strideR, int res_rows,
                           const void* pos_y, const void* pos_x, long ld_pos, int pos_h, int pos_w, int pos_f16,
                           int act, int M, int N, int K, int batch, hipStream_t stream) {
  ZH_CHECK_ARG(A && W && C, "zh_gemm_f16: null operand");
  ZH_CHECK_ARG(M > 0 && N > 0 && K > 0 && batch > 0, "zh_gemm_f16: bad shape M=%d N=%d K=%d batch=%d", M, N, K, batch);
  ZH_CHECK_ARG(K % 64 == 0, "zh_gemm_f16: K=%d must be a multiple of 64", K);
  ZH_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && strideA % 8 == 0 && strideW % 8 == 0,
               "zh_gemm_f16: lda/ldw/strides must be multiples of 8 halves (16-byte rows)");
  ZH_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0, "zh_gemm_f16: A/W must be 16-byte aligned");
  ZH_CHECK_ARG(act >= 0 && act <= 4, "zh_gemm_f16: bad activation %d", act);
  // the kernel addresses an operand row as base + (32-bit element offset): keep one batch item's A / W below 2^32 elements
  ZH_CHECK_ARG((long)(M - 1) * lda + K <= 0xFFFFFFFFL && (long)(N - 1) * ldw + K <= 0xFFFFFFFFL,
               "zh_gemm_f16: an operand exceeds 2^32 elements per batch item (M=%d lda=%ld N=%d ldw=%ld)", M, lda, N, ldw);
  ZH_CHECK_ARG(!residual || res_rows > 0, "zh_gemm_f16: residual needs res_rows > 0");
  ZH_CHECK_ARG(zh_pos_tables_ok(pos_y, pos_x, ld_pos, pos_h, pos_w, N), "zh_gemm_f16: pos tables need both pointers 16-byte aligned, "
               "pos_h, pos_w > 0, ld_pos %% 8 == 0 and N %% 4 == 0");
  GemmArgs p;
  p.A = (const half_t*)A; p.lda = lda; p.sA = strideA;
  p.W = (const half_t*)W; p.ldw = ldw; p.sW = strideW;
  p.C = C; p.ldc = ldc; p.sC = strideC;
  p.planeA = p.planeW = p.planeC = 0; p.out_scale = 1.0f;
  p.bias = bias; p.R = residual; p.ldr = ldr; p.sR = strideR; p.res_rows = res_rows;
  p.pos_y = pos_y; p.pos_x = pos_x; p.ld_pos = ld_pos; p.pos_hw = pos_h * pos_w; p.pos_w = pos_w; p.pos_f16 = pos_f16;
  p.M = M; p.N = N; p.K = K; p.act = act; p.nbm = p.nbn = 0;
  // super-tile height: 3..8 measure within 1-3 % of each other on the model (4: least fabric traffic, 134 vs 141 MB per launch,
  // and the best step), 14: -1 %, 16 / 32 lose 13 / 36 % on 8192^3 (tools/gemm_vs_blaslt.py) — the order in which an XCD's 32
  // resident tiles share panels matters
  const GemmDevOverrides& dev = gemm_dev_overrides();
  p.group_m = dev.group_m;
#ifdef ZH_GEMM_PROBE
  p.probe = g_probe;
#endif
  const int esz = out_f16 ? 2 : 4;
  p.vec_ok = (N % 4 == 0) && (ldc % 4 == 0) && (strideC % 4 == 0) && (((uintptr_t)C & (4 * esz - 1)) == 0) &&
             (!bias || ((uintptr_t)bias & 15) == 0) &&
             (!residual || (ldr % 4 == 0 && strideR % 4 == 0 && ((uintptr_t)residual & 15) == 0));
  ZH_CHECK_ARG((long)zh_cdiv(M, 64) * zh_cdiv(N, 64) * batch < (1L << 31), "zh_gemm_f16: grid too large");
  const double c256 = tiling_cost(M, N, batch, 256, 256, 1, 1.0);
  const double c192 = tiling_cost(M, N, batch, 256, 192, 1, 0.95);
  const double c128 = tiling_cost(M, N, batch, 128, 128, 2, 0.8);
  int pick = (c128 < c256 && c128 < c192) ? 128 : (c192 < c256 ? 192 : 256);
  // (a 128x64 tile for the decoder's M = B*Q GEMMs measured no better than 128x128: they are slice-latency bound;
  //  it stays reachable through ZH_GEMM_TILE=64 for experiments)
  // few-tile GEMMs (batch-1 inference: M = 442 tokens -> 24 tiles of 128x128 on 256 CUs): 64x64 tiles on an 8-deep
  // ring put 4x the CUs to work; measured 32 -> 16 us on the 442x768x3072 MLP projection (tools/gemm_small.py)
  const long t128 = (long)zh_cdiv(M, 128) * zh_cdiv(N, 128) * batch;
  if (t128 <= 96) pick = 3064;
  // narrow outputs that do not fill the chip with 128 x 128 tiles (decoder M = B*Q rows x N = 768): 128 x 64 tiles put twice the
  // blocks to work; measured 19.6 -> 16.5 us (K = 768) and 32.7 -> 28.8 us (K = 2048) with the residual epilogue, while
  // N = 1536 / 2048 keep 128 x 128 (tools/gemm_dec_tiles.py)
  else if (pick == 128 && t128 <= 256 && N <= 768) pick = 64;
#ifndef ZH_F16_ROUND4_SMALL_TILES   // developer A/B (tools/build_variant_lib.sh): the one-round tiles as selected until round 5
  // Round 5: one-round tiles on 64-k slices (128-B row pieces) with several slices in flight (gemm_kernel.h K64 / gemm_k64_plain; the
  // finding of the split-pair tiles, tools/gemm_small_bench.py k64f): 64 x 64 on seven slots instead of eight 32-k ones (c_proj at one
  // image 18.6 -> 15.1 us, at 442 tokens 17.4 -> 13.7); 128 x 96 on five where it makes ONE round (the decoder's 3200 x 768 x 768
  // 14.7 -> 13.1 / 12.3 -> 10.4 us, linear2 (K = 2048) 26.7 -> 22.1, QKV at one image 11.7 -> 10.7); 128 x 128 on five where that is one
  // round (c_fc at one image 14.2 -> 13.7).  (128 x 64 on six slots: slower than every form it was compared with — not kept.)
  if (pick == 3064) pick = 7032;
  else if ((pick == 64 || pick == 128) && N % 96 == 0 && (long)zh_cdiv(M, 128) * (N / 96) * batch <= 256) pick = 7096;
  else if (pick == 128 && t128 <= 256) pick = 7128;
#endif
  const int forced = dev.tile ? dev.tile : (M <= 4096 ? dev.tile_small : 0);
  if (forced) {
    static const int known[] = {64, 128, 192, 256, 2064, 2128, 3064, 7032, 7096, 7128};
    bool okc = false;
    for (int k : known) okc |= (k == forced);
    ZH_CHECK_ARG(okc, "zh_gemm_f16: ZH_GEMM_TILE(_SMALL)=%d is not a tile code (64|128|192|256|2064|2128|3064|7032|7096|7128)", forced);
    pick = forced;
  }
  // 16-byte row stores need 16-B aligned rows; an f16 residual is not supported (none on the hot path)
  const bool wide_ok = p.vec_ok && (((uintptr_t)C & 15) == 0) && ((ldc * esz) % 16 == 0) && ((strideC * esz) % 16 == 0) &&
                       ((N * esz) % 16 == 0) && !(out_f16 && residual);
  // Tail peel (the f16x3 dispatcher's, gemm_x3.hip): a big-tile GEMM a few tiles over whole rounds of the 256-CU chip pays a full
  // round for them (config 5's out_proj / c_proj at fp16, 147712 x 1024: 2308 tiles of 256 x 256 = 9 rounds + 4 tiles).  When the
  // surplus is at most a quarter round, the last m-tile rows that hold it go into a second call on the row range [M', M) (its own
  // tile choice; the K order of a tile does not depend on the tile shape: bitwise one launch).  Plain pointer offsets only: no
  // batch, no pos tables, a residual with a row of its own per output row.
  if ((pick == 256 || pick == 192) && wide_ok && batch == 1 && !pos_y && (!residual || res_rows >= M) && !forced) {
    const long BNp = pick == 256 ? 256 : 192;
    const long nbm = zh_cdiv(M, 256), nbn = zh_cdiv(N, BNp), tiles = nbm * nbn, rem = tiles % 256;
    if (tiles > 4 * 256 && rem > 0 && rem <= 64) {
      const long r = (rem + nbn - 1) / nbn;                       // m-tile rows to peel
      const long M1 = (nbm - r) * 256;                            // rows that stay: whole tiles, whole rounds (or just under)
      if (r * nbn <= 64 && M1 > 0 && M1 < M) {
        int rc = zh_gemm_f16(A, lda, strideA, W, ldw, strideW, C, ldc, strideC, out_f16, bias, residual, ldr, strideR, residual ? res_rows : 0,
                             pos_y, pos_x, ld_pos, pos_h, pos_w, pos_f16, act, (int)M1, N, K, 1, stream);
        if (rc != ZH_OK) return rc;
        return zh_gemm_f16((const char*)A + M1 * lda * 2, lda, strideA, W, ldw, strideW, (char*)C + M1 * ldc * esz, ldc, strideC, out_f16, bias,
                           residual ? residual + M1 * ldr : nullptr, ldr, strideR, residual ? (int)(res_rows - M1) : 0,
                           pos_y, pos_x, ld_pos, pos_h, pos_w, pos_f16, act, (int)(M - M1), N, K, 1, stream);
      }
    }
  }
  bool ok;
  if (!p.vec_ok) ok = launch_gemm<2, 2, 4, 4, 4, 0>(p, batch, out_f16, stream);      // scalar-store fallback: small tile only
  else if (!wide_ok) ok = launch_gemm<2, 2, 4, 4, 4, 1>(p, batch, out_f16, stream);  // direct 8/16-B stores
  else if (pick == 2128) ok = launch_gemm<2, 2, 4, 4, 8, 2>(p, batch, out_f16, stream);   // 128 x 128, 8-deep ring
  else if (pick == 2064) ok = launch_gemm<2, 2, 4, 2, 8, 2>(p, batch, out_f16, stream);   // 128 x 64, 8-deep ring
  else if (pick == 3064) ok = launch_gemm<2, 2, 2, 2, 8, 2>(p, batch, out_f16, stream);   // 64 x 64, 8-deep ring
  else if (pick == 7032) ok = launch_gemm<2, 2, 2, 2, 7, 2>(p, batch, out_f16, stream);   // 64-k slices (gemm_k64_plain): 64 x 64, seven slots
  else if (pick == 7096) ok = launch_gemm<2, 2, 4, 3, 5, 2>(p, batch, out_f16, stream);   // ... 128 x 96, five
  else if (pick == 7128) ok = launch_gemm<2, 2, 4, 4, 5, 2>(p, batch, out_f16, stream);   // ... 128 x 128, five
  else if (pick == 64) ok = launch_gemm<2, 2, 4, 2, 4, 2>(p, batch, out_f16, stream);   // 128 x 64
  else if (pick == 128) ok = launch_gemm<2, 2, 4, 4, 4, 2>(p, batch, out_f16, stream);
  else if (pick == 192) ok = launch_gemm<2, 4, 8, 3, 4, 2>(p, batch, out_f16, stream);
  else ok = launch_gemm<2, 4, 8, 4, 4, 2>(p, batch, out_f16, stream);
  ZH_CHECK_ARG(ok, "zh_gemm_f16: (out_f16=%d, act=%d) is not an instantiated epilogue (f32: none|sigmoid; f16: none|quickgelu|relu|gelu_erf)",
               out_f16, act);
  ZH_CHECK_LAUNCH("zh_gemm_f16");
  return ZH_OK;
}
