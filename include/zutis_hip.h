/* libzutis_hip — C ABI of the MI355X (gfx950) kernels behind the ZUTIS dense-prediction hot path.
 *
 * The reference (NoelShin/zutis) has no FFI layer: its boundary is the Python nn.Module call surface
 * (SURVEY.md §8b).  This header is the boundary *underneath* this build's Python mirror of that surface
 * (zutis_amd/dropin/networks/zutis.py ...): every entry point replaces one stock-op sequence of the reference,
 * cited per function as file:line relative to the reference root.
 *
 * Conventions
 *   - extern "C"; every entry returns int: 0 = ok, <0 = error (ZH_ERR_*); zh_last_error() gives thread-local text.
 *   - All pointers are raw DEVICE pointers owned by the caller (PyTorch tensors); the library never allocates,
 *     frees or synchronises.  Scratch is passed in (void* workspace, size_t bytes) with a *_workspace_size query.
 *   - Last argument: the hipStream_t to launch on (pass torch.cuda.current_stream().cuda_stream).
 *   - Layout: contiguous row-major, tokens channels-last [B, T, D].  "f16" = IEEE half; fp32 accumulate everywhere.
 *   - Split pairs (the reference-equivalent "f16x3" precision): a tensor x stored as two fp16 planes, hi = f16(x) at the
 *     pointer and lo = f16(x - hi) `lo_plane` ELEMENTS further on (22 significand bits).  Producers take a `lo_plane`
 *     argument (0 = write the plain fp16 tensor only; the hi plane alone IS that tensor); zh_gemm_f16x3 and the split-pair
 *     form of zh_attention_f16 consume them.
 *   - Re-entrant.  Process-wide state: none on the product path; the developer entry zh_dev_set_gemm_overrides() (forced GEMM tile /
 *     super-tile height, used by tests and tools only) is process-wide by design, and zh_last_error() text is thread-local.
 */
#ifndef ZUTIS_HIP_H
#define ZUTIS_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* zh_stream_t; /* == hipStream_t */

#define ZH_OK 0
#define ZH_ERR_ARG (-1)
#define ZH_ERR_HIP (-2)
#define ZH_ERR_WORKSPACE (-3)

/* GEMM epilogue activations */
#define ZH_ACT_NONE 0
#define ZH_ACT_QUICKGELU 1 /* x*sigmoid(1.702x)  networks/clip_arch.py:295-297 */
#define ZH_ACT_RELU 2      /* networks/zutis.py:546-549; networks/transformer.py:289 */
#define ZH_ACT_SIGMOID 3   /* networks/zutis.py:209 */
#define ZH_ACT_GELU_ERF 4  /* nn.GELU, networks/selfmask/vision_transformer.py:79 */

/* ABI version: bumped whenever an entry point's signature changes.  zh_version() returns the value the library was BUILT
 * with; a binding compiled / written against this header must refuse a library that reports another one (zutis_amd/_lib.py
 * does) — ctypes cannot see a changed argument list. */
#define ZH_ABI_VERSION 225 /* 225: zh_dev_set_gemm_persist; 224: zero_word of zh_mask_nms; 223: zh_mask_rle_fused_kept; 222: zh_mask_rle_kept; 221: packed_capacity of zh_mask_runs_kept (the kept masks' transitions as one list), packed form of zh_rle_from_transitions_host; 220: flags argument of zh_gemm_f16x3 (ZH_GEMM_FIXED_K_ORDER); 219: workspace of zh_mask_runs / zh_mask_runs_kept (two-launch run extraction); 218: status word of the LayerNorm family, f16_scale of the unit-norm producers; 217: zh_mask_runs_kept, range_flag / packed arguments of zh_instance_mask_stats / zh_mask_nms; 216: zh_sum_layernorm_f32, few-row kernel behind zh_gemm_f16x3; 215: zh_rle_from_transitions_host; 214: zh_gemm_f16x3 accepts planeW = 0 (fp16-valued weight: two products); 213: workspace argument of zh_masked_mean_tokens; 212: zh_attention_f16_splitk; 211: zh_dev_set_gemm_overrides; 210: pos_y / pos_x tables on zh_gemm_f16 / zh_gemm_f16x3 */
int zh_version(void);
const char* zh_arch(void);
const char* zh_last_error(void);

/* DEVELOPER entry (tests / tools, not part of the reference's surface): force the GEMM tile variant for the calls that follow.
 * group_m = super-tile height (0 = default 4); tile = tile code for zh_gemm_f16 (64|128|192|256|2064|2128|3064; 7032 / 7096 / 7128 = the
 * 64 x 64 / 128 x 96 / 128 x 128 tiles on 64-k slices) and zh_gemm_f16x3 (64|96|192|256|448|512|3064|1288; 3066 = 64 x 64 on 32-k slices,
 * 6464 / 6496 = 128 x 64 / 128 x 96 on 64-k slices in three / two whole slots, 7096 / 7128 = 128 x 96 / 128 x 128 on the circular piece ring;
 * 5122 / 5124 = the planeW = 0 form of 512 on two slots / as 2 x 4 waves of 128 x 64, 4484 = of 448 as 4 x 2 waves of 48 x 128), 0 = the cost model's choice; tile_small = the same, applied to M <= 4096 only.  Process-wide;
 * the initial values come from ZH_GEMM_GROUP_M / ZH_GEMM_TILE / ZH_GEMM_TILE_SMALL, read once. */
int zh_dev_set_gemm_overrides(int group_m, int tile, int tile_small);

/* DEVELOPER entry: the big plain-fp16 GEMM tiles (256 x 256 / 256 x 192) are persistent — `workgroups` of them walk a launch's tiles and
 * request the next tile's first K slices under the current tile's epilogue (csrc/gemm_kernel.h PERS; bitwise the one-workgroup-per-
 * tile results).  Default (-1): the current device's CU count rounded down to a multiple of 8, resolved at the first launch (256 on an
 * MI355X in SPX mode; ZH_GEMM_PERSIST, validated the same way, read once); 0 = one workgroup per tile; any multiple of 8 forces that
 * many (tests use 8 to make small problems walk several tiles per workgroup).  Process-wide: the library's one mutable word that
 * steers product launches, written only here. */
int zh_dev_set_gemm_persist(int workgroups);

/* C[b][m][n] = act(sum_k A[b][m][k]*W[b][n][k] + bias[n] + pos[m][n]) + residual[b][m % res_rows][n]
 * A [M,K] f16 (lda), W [N,K] f16 (ldw) — torch Linear layout; C f32 or f16 (out_f16); bias/residual f32 or NULL.
 * K % 64 == 0; lda/ldw % 8 == 0.  residual may alias C (in-place x += ...).
 * pos (optional, pos_y and pos_x both NULL = none): output rows are pixels, m = image * pos_h*pos_w + y * pos_w + x, and
 * pos[m][n] = pos_y[y][n] + pos_x[x][n] with tables [pos_h, ld_pos] / [pos_w, ld_pos], fp32 or (pos_f16) fp16, ld_pos % 8 == 0,
 * N % 4 == 0.  This is the
 * `pos` term of the decoder's key projection of `memory + pos` (transformer.py:281-283): the sine PE is [py(y) | px(x)]
 * (positional_embedding.py:47-52), so pos @ Wk^T separates into two small input-independent tables and `memory + pos` is
 * never materialised; the accumulators start from the table values (read under the operand prefetch). */
int zh_gemm_f16(const void* A, long lda, long strideA, const void* W, long ldw, long strideW,
                void* C, long ldc, long strideC, int out_f16,
                const float* bias, const float* residual, long ldr, long strideR, int res_rows,
                const void* pos_y, const void* pos_x, long ld_pos, int pos_h, int pos_w, int pos_f16,
                int act, int M, int N, int K, int batch, zh_stream_t stream);

/* The same contraction at the reference's precision (the reference computes every Linear / einsum in fp32:
 * clip_arch.py:286-292 keeps LayerNorm fp32, zutis.py:55 casts the CLIP weights to fp32).  A and W are split pairs
 * (planeA / planeW = element offset of the lo plane); the kernel accumulates Ah.Wh + Ah.Wl + Al.Wh in fp32 (dropped
 * term: 2^-22 relative) and multiplies the accumulator by out_scale (weights are packed as W * 2^s, out_scale = 2^-s,
 * so that lo planes stay normal fp16 numbers) before the bias.  out_kind: 0 = f32 (residual allowed), 1 = f16,
 * 2 = split pair (lo plane at C + planeC).
 * planeW = 0: W has NO lo plane — every value of W * 2^s is an fp16 number, as for the released CLIP towers, whose weights the
 * reference's own constructor rounds to fp16 (convert_weights, clip_arch.py:566-587,625) before zutis.py:55 / encode_image use
 * them: the kernel then issues Ah.Wh + Al.Wh only, bit-identical to the three-product form on a zero lo plane, at 2/3 of the
 * MFMA work ("f16x2").  A is always a split pair.
 * flags: ZH_GEMM_FIXED_K_ORDER (1) = the caller compares results of calls with different M / N bit for bit (sharded retrieval): only
 * the LDS-ring kernels, whose K order is the same for every tile shape, are used; 0 = few-row problems (M <= 128, a small grid) take the
 * few-row kernel (gemm_skinny.h: K split over the waves of a workgroup — a different, equally fp32-class summation order). */
#define ZH_GEMM_FIXED_K_ORDER 1
int zh_gemm_f16x3(const void* A, long lda, long strideA, long planeA, const void* W, long ldw, long strideW, long planeW,
                  void* C, long ldc, long strideC, long planeC, int out_kind, float out_scale,
                  const float* bias, const float* residual, long ldr, long strideR, int res_rows,
                  const void* pos_y, const void* pos_x, long ld_pos, int pos_h, int pos_w, int pos_f16,
                  int act, int M, int N, int K, int batch, int flags, zh_stream_t stream);

/* Flash attention: O = softmax(scale * Q K^T) V per (image, head); Q [Tq, heads*dh] rows with stride ldq, etc.
 * f16 in/out, fp32 softmax/accumulate; head_dim in {64, 96}.
 * planeQ / planeK / planeV != 0 (all or none): Q, K, V are split pairs; scores are Kh.Qh + Kl.Qh + Kh.Ql and the
 * probabilities are split in registers, O += Vh.Ph + Vl.Ph + Vh.Pl (fp32-class).  planeO != 0: O is written as a split pair.
 * Replaces nn.MultiheadAttention core at clip_arch.py:314-316, transformer.py:272-286,
 * selfmask/vision_transformer.py:110-133 (which materialises [B,heads,T,T]). */
/* The same attention with the KEYS split over `ksplit` (2..64) workgroups per (image, head, 128-query block) and one merge launch:
 * for few queries against many keys (the decoder's cross-attention, transformer.py:281-286: 100 queries x 1764 keys).  Partials
 * (unnormalised fp32 O, running max, row sum) go through `workspace` (zh_attention_splitk_workspace_size bytes, 16-byte aligned). */
size_t zh_attention_splitk_workspace_size(int batch, int heads, int Tq, int head_dim, int ksplit);
int zh_attention_f16_splitk(const void* Q, long ldq, long strideQ, const void* K, long ldk, long strideK,
                            const void* V, long ldv, long strideV, void* O, long ldo, long strideO,
                            int batch, int heads, int Tq, int Tk, int head_dim, float scale,
                            long planeQ, long planeK, long planeV, long planeO, int ksplit, void* workspace, size_t workspace_bytes,
                            zh_stream_t stream);
int zh_attention_f16(const void* Q, long ldq, long strideQ, const void* K, long ldk, long strideK,
                     const void* V, long ldv, long strideV, void* O, long ldo, long strideO,
                     int batch, int heads, int Tq, int Tk, int head_dim, float scale,
                     long planeQ, long planeK, long planeV, long planeO, zh_stream_t stream);

/* The same under CLIP's causal mask (build_attention_mask, clip_arch.py:525-531: -inf above the diagonal), Tq = Tk = T:
 * the text tower's resblocks (clip_arch.py:534-541). */
int zh_attention_causal_f16(const void* Q, long ldq, long strideQ, const void* K, long ldk, long strideK,
                            const void* V, long ldv, long strideV, void* O, long ldo, long strideO,
                            int batch, int heads, int T, int head_dim, float scale,
                            long planeQ, long planeK, long planeV, long planeO, zh_stream_t stream);

/* Text tower glue.  x = token_embedding(text) + positional_embedding (clip_arch.py:535-537): tokens int64 [n,ctx],
 * table f32 [vocab,D], pos f32 [ctx,D] -> out f32 [n*ctx, D]. */
int zh_embed_tokens_f32(const long long* tokens, const float* table, const float* pos, float* out, long n, int ctx, int D,
                        int vocab, zh_stream_t stream);
/* x[arange(n), text.argmax(-1)] (clip_arch.py:545): x f32 [n*ctx, D] -> out f32 [n, D]; first maximum on ties. */
int zh_eot_rows_f32(const long long* tokens, const float* x, float* out, long n, int ctx, int D, zh_stream_t stream);
/* Prompt ensembling tail (utils/extract_text_embeddings.py:110-112): x f32 [groups,T,E] unit-norm rows ->
 * out[g] = mean_t x[g,t] / ||mean_t x[g,t]||. */
int zh_group_mean_l2norm(const float* x, float* out, int groups, int T, int E, zh_stream_t stream);

/* Row LayerNorm (biased variance, eps inside sqrt): clip_arch.py:286-292, transformer.py:249-251, 140-150.
 * in_row(r) = (r / in_group_rows)*in_group_stride + in_offset + r % in_group_rows   (drops the cls token for ln_post,
 * clip_arch.py:403-404); out_row(r) uses the same form (stacks decoder layers as [B,L,Q,D], transformer.py:140-150).
 * Outputs (any may be NULL): y f32, y f16, (y + add[r % add_rows]) f16 / f32
 * (query_pos add, transformer.py:268,277). gamma/beta both NULL => no affine. */
int zh_layernorm_f32(const float* x, long in_group_rows, long in_group_stride, long in_offset,
                     long out_group_rows, long out_group_stride, long out_offset,
                     const float* gamma, const float* beta, float eps,
                     float* out_f32, void* out_f16, void* out_f16_plus, float* out_f32_plus,
                     const float* add, int add_rows, int rows, int D, long lo_plane, int* status, zh_stream_t stream);
/* status (here, zh_sum_layernorm_f32, zh_global_ln_l2; may be NULL): a device word into which ZH_STATUS_NONFINITE (2) is OR-ed when a row's
 * variance is inf / NaN.  The split-pair (f16x3) operands of this library hold |x| < 65504: beyond it hi = inf and every later product is
 * a NaN that reaches the next LayerNorm of the path — checking there costs one compare per row.  The host reads the word once per
 * forward (at its next synchronisation) and raises; bit 0 of the same word is zh_instance_mask_stats' range flag. */
#define ZH_STATUS_RANGE 1
#define ZH_STATUS_NONFINITE 2

/* Split-K combine + bias + residual + LayerNorm (+ a second, chained LayerNorm) in one pass over each row (round 4):
 *   x = ((parts[0] + ... + parts[n_parts-1]) + bias) + residual      parts f32 [n_parts][rows, D] at part_stride, in plane order
 *   out_sum[r] = x                                                  (may alias residual: `x = x + attn(...)`, clip_arch.py:318-321)
 *   y = LN(x; gamma, beta, eps)   -> out_f32 / out_f16 at out_row(r) = (r / out_group_rows) * out_group_stride + out_offset + r % out_group_rows,
 *                                    not written for r % out_group_rows == 0 when skip_first_in_group (ln_post drops cls, clip_arch.py:403-404)
 *   z = LN(y; gamma2, beta2, eps2) -> out2_* at out2_row(r)         (norm3 -> decoder.norm, transformer.py:140-150,291)
 * Replaces the fp32 epilogue of the N = D GEMMs whose K was split over workgroups (zh_gemm_f16x3 with batch = S over K slabs) and the
 * LayerNorm launch that followed them.  gamma NULL: only out_sum.  Any output may be NULL. */
int zh_sum_layernorm_f32(const float* parts, int n_parts, long part_stride, const float* bias, const float* residual,
                         float* out_sum, const float* gamma, const float* beta, float eps,
                         float* out_f32, void* out_f16, long lo_plane,
                         long out_group_rows, long out_group_stride, long out_offset, int skip_first_in_group,
                         const float* gamma2, const float* beta2, float eps2, float* out2_f32, void* out2_f16, long lo_plane2,
                         long out2_group_rows, long out2_group_stride, long out2_offset,
                         int rows, int D, int* status, zh_stream_t stream);

/* cat(class_embedding, patch_emb) + pos_embed, then ln_pre: clip_arch.py:384-397.  out [B,T,D] f32.
 * gamma = beta = NULL: no LayerNorm (DINO ViT prepare_tokens, selfmask/vision_transformer.py:269-281). */
int zh_assemble_tokens_ln(const float* patch_emb, const float* class_embedding, const float* pos_embed,
                          const float* gamma, const float* beta, float eps, float* out,
                          int B, int T, int D, zh_stream_t stream);

/* x / (||x||_2 + eps) per row: queries (eps = 0) zutis.py:515; averaged tokens (eps = 1e-7) zutis.py:413. */
int zh_l2norm_rows(const float* x, float* out_f32, void* out_f16, float eps, int rows, int D, long lo_plane, float f16_scale, zh_stream_t stream);
/* f16_scale (zh_l2norm_rows, zh_global_ln_l2, zh_cast_f32_f16; a power of two, 1 = none): the fp16 / split-pair copy holds y * f16_scale; its
 * consumer multiplies its accumulator by 1 / f16_scale (zh_gemm_f16x3 out_scale).  Unit-norm rows (|y| ~ 0.04) are stored times 2^10 so that
 * the lo half of the pair is a normal fp16 number (22 significant bits instead of ~19); fp32 outputs are never scaled. */

/* F.layer_norm over the whole (h,w,c) volume per image (no affine) then x/(||x||_c + l2_eps): zutis.py:320-322. */
size_t zh_global_ln_l2_workspace_size(int B, int M, int C);
int zh_global_ln_l2(const float* x, float* out_f32, void* out_f16, float eps, float l2_eps,
                    int B, int M, int C, void* workspace, size_t workspace_bytes, long lo_plane, float f16_scale, int* status,
                    zh_stream_t stream);

/* im2col of the stride==kernel patch conv (pure re-index): clip_arch.py:340,378; selfmask/vision_transformer.py:182.
 * out f16 [B*gh*gw, Kpad], k = c*p*p + i*p + j, zero padded.  pad_to_patch = 0: gh = floor((H-p)/p)+1 (CLIP, trailing
 * pixels dropped); 1: gh = ceil(H/p) with zero pixels (make_input_divisible, selfmask/vision_transformer.py:260-267). */
int zh_im2col_f16(const float* x, void* out, int B, int Cin, int H, int W, int patch, int Kpad, int pad_to_patch,
                  long lo_plane, zh_stream_t stream);

/* Bicubic positional-embedding resample: clip_arch.py:356-374 (scale = float(1/((h+0.1)/g))) and
 * selfmask/vision_transformer.py:377-401 (scale = g/h).  pos [has_cls + g*g, D] -> out [has_cls + h*w, D]. */
int zh_posembed_bicubic(const float* pos, float* out, int grid, int h, int w, int D, float scale_h, float scale_w,
                        int has_cls, zh_stream_t stream);

/* F.interpolate(scale_factor=2, bilinear) on channels-last tokens: zutis.py:491-495.  relu != 0: max(., 0) after the
 * interpolation — the engine applies the (linear) first ffn1 layer and the text-space projection to the h*w tokens and
 * upsamples their outputs (a linear map commutes with the interpolation, whose weights sum to 1), zutis.py:491-503,319. */
int zh_upsample2x_bilinear_cl(const float* x, float* out_f32, void* out_f16, int B, int h, int w, int D, long lo_plane,
                              int relu, zh_stream_t stream);

/* PositionEmbeddingSine(normalize=True): positional_embedding.py:29-52 -> [h*w, D] channels-last. */
int zh_sine_pe(float* out, int h, int w, int D, float temperature, zh_stream_t stream);

/* memory + pos (transformer.py:281): out f16 = a f16 + add f32[r % add_rows].  a_lo_plane != 0: a is a split pair. */
int zh_add_rowperiodic_f16(const void* a, const float* add, void* out, long rows, int D, int add_rows, long a_lo_plane,
                           long lo_plane, zh_stream_t stream);

/* x[i] = value (tgt = zeros_like(queries), zutis.py:164) — a kernel so that it can be part of a launch plan. */
int zh_fill_f32(float* x, float value, long n, zh_stream_t stream);

/* f32 -> f16 (optionally + add[r % add_rows]). */
int zh_cast_f32_f16(const float* x, const float* add, int add_rows, void* out, long rows, int D, long lo_plane, float f16_scale,
                    zh_stream_t stream);

/* argmax_c(F.interpolate(logits, size=(H,W), bilinear)) fused, bit-identical to ATen incl. ties: zutis.py:366-372.
 * logits_lo f32 [B,n,h,w] -> labels int64 [B,H,W].  scale_* = float32(in)/float32(out) computed by the host. */
int zh_upsample_argmax(const float* logits_lo, long long* labels, int B, int n, int h, int w, int H, int W,
                       float scale_h, float scale_w, zh_stream_t stream);

/* F.interpolate(x, size, bilinear) on `planes` NCHW planes (return_logits zutis.py:368-371; masks zutis.py:422-423);
 * optional mask_u8 = value > threshold. */
int zh_upsample_bilinear_nchw(const float* x, float* out, unsigned char* mask_u8, float threshold, long planes,
                              int h, int w, int H, int W, float scale_h, float scale_w, zh_stream_t stream);

/* SelfMask inference tail, selfmask.py:207-221: per image the query with the largest objectness logit (first maximum),
 * its mask plane [h,w] bilinearly resampled (scale_* as for zh_upsample_bilinear_nchw; H,W = cropped output) and
 * thresholded -> out_u8 [B,H,W], index int64 [B].  objectness f32 [B,Q] (logits), masks f32 [B,Q,h,w]. */
int zh_select_upsample_mask(const float* objectness, const float* masks, unsigned char* out_u8, long long* index, int B, int Q,
                            int h, int w, int H, int W, float scale_h, float scale_w, float threshold, zh_stream_t stream);

/* F.interpolate(mask[None,None], size=(H,W), mode="nearest") on a u8 mask: datasets/index_dataset.py:215
 * (restore the original resolution of a pseudo-mask).  scale_* = float32(in)/float32(out). */
int zh_resize_nearest_u8(const unsigned char* x, unsigned char* out, int h, int w, int H, int W, float scale_h, float scale_w,
                         zh_stream_t stream);

/* RunningScore._fast_hist: utils/running_score.py:11-16.  hist_accum int64 [n*n] += bincount(n*gt+pred), 0<=gt<n. */
int zh_confusion_hist(const long long* label_true, const long long* label_pred, long long* hist_accum, long total,
                      int n_class, zh_stream_t stream);

/* ---- instance prediction, zutis.py:374-420 ---- */
/* binary = p > threshold; size = sum(binary); confidence = sum(p*binary)/(size+1e-7)   (zutis.py:390-397).
 * mask_proposals f32: image b at + b*stride_image, [Q, M] contiguous inside (last decoder layer slice). */
int zh_instance_mask_stats(const float* mask_proposals, long stride_image, float threshold, int B, int Q, int M,
                           float* sizes, float* confidence, unsigned char* binary, int* range_flag, zh_stream_t stream);
/* range_flag (may be NULL; zero it first): bit 0 is OR-ed in when a proposal is outside [0, 1] or a NaN — the reference's range asserts
 * (zutis.py:385-386) without a reduction and a device -> host copy of their own. */
/* avg[b,q,:] = sum_m binary[b,q,m]*tokens[b,m,:] / (size+1e-7)  (zutis.py:404-406; the reference materialises
 * B x Q x hw x E).  Pixels are processed in chunks of 128 by separate workgroups; per-chunk partial sums go through `workspace`
 * (zh_masked_mean_workspace_size bytes) and are added in chunk order. */
size_t zh_masked_mean_workspace_size(int B, int Q, int M, int E);
int zh_masked_mean_tokens(const float* tokens, const unsigned char* binary, const float* sizes, float* avg,
                          int B, int Q, int M, int E, void* workspace, size_t workspace_bytes, zh_stream_t stream);
/* category = argmax_n sigmoid(T * text_n . avg/(||avg||+1e-7)); score = confidence * max  (zutis.py:409-420). */
int zh_instance_classify(const float* avg, const float* text, const float* confidence, float temperature,
                         int rows, int n_classes, int E, long long* category, float* score, zh_stream_t stream);
/* pairwise |a&b|, |a|b| of n {0,1} u8 masks (bit-packed popcount): utils/iou.py:6-37 inside the NMS loop
 * zutis.py:245-278. */
size_t zh_mask_iou_workspace_size(int n, long pixels);
int zh_mask_iou_counts(const unsigned char* masks, int n, long pixels, int* inter, int* uni,
                       void* workspace, size_t workspace_bytes, zh_stream_t stream);

/* ---- bilateral solver (SelfMask refinement), utils/bilateral_solver.py:40-195; float64 ---- */
/* convert_tensor_to_pil_image, utils/utils.py:261-273: fp32 x*std+mean, *255, clip, TRUNCATE.  x f32 [3,H,W] (device)
 * -> rgb u8 [H,W,3] (device).  mean3/std3 are HOST float[3]. */
int zh_denormalize_u8(const float* x, unsigned char* rgb, int H, int W, const float* mean3, const float* std3,
                      zh_stream_t stream);
/* per-pixel lattice coordinates (x/ss, y/ss, Y/sl, U/sc, V/sc), bilateral_solver.py:42-50 -> int32 [H*W,5] (parity probe). */
int zh_bgrid_coords(const unsigned char* rgb, int H, int W, double sigma_spatial, double sigma_luma, double sigma_chroma,
                    int* coords, zh_stream_t stream);
/* BilateralGrid + bistochastize + BilateralSolver.solve for one channel: bilateral_solver.py:58-149 with the
 * constants of bilateral_solver_output (:162-175: confidence 0.999, lam 256, A_diag_min 1e-5, cg_tol 1e-5, maxiter 25)
 * passed by the caller.  rgb u8 [H,W,3]; target u8 [H,W] or f64 [H,W] (exactly one non-NULL); out_soft f64 [H,W];
 * stats int32 [2] = {nvertices, cg iterations} (device, may be NULL); n_out/m_out f64 [H*W] debug copies of the
 * bistochastisation vectors (may be NULL). */
size_t zh_bilateral_workspace_size(int H, int W, double sigma_spatial, double sigma_luma, double sigma_chroma);
int zh_bilateral_solve(const unsigned char* rgb, const unsigned char* target_u8, const double* target_f64, int H, int W,
                       double sigma_spatial, double sigma_luma, double sigma_chroma, double confidence, double lam,
                       double a_diag_min, double cg_tol, int cg_maxiter, double* out_soft, int* stats,
                       double* n_out, double* m_out, void* workspace, size_t workspace_bytes, zh_stream_t stream);

/* The same for B images of one size in ONE sequence of launches (blockIdx.y = image): rgb u8 [B,H,W,3], target [B,H,W],
 * out_soft f64 [B,H,W], stats int32 [B,2], n_out / m_out f64 [B,H*W]; workspace = B * zh_bilateral_workspace_size(...) bytes.
 * A solve is ~110 small launches (grid build, 11 bistochastisation steps, 3 per CG iteration): batching shares them —
 * the pseudo-label driver's batched mode (datasets/index_dataset.py:189-204 runs batch 1). */
int zh_bilateral_solve_batch(const unsigned char* rgb, const unsigned char* target_u8, const double* target_f64, int B, int H, int W,
                             double sigma_spatial, double sigma_luma, double sigma_chroma, double confidence, double lam,
                             double a_diag_min, double cg_tol, int cg_maxiter, double* out_soft, int* stats,
                             double* n_out, double* m_out, void* workspace, size_t workspace_bytes, zh_stream_t stream);

/* output_solver > 0.5 on the device (networks/selfmask/selfmask.py:231): x f64 [n] -> out u8 {0,1} [n]. */
int zh_threshold_f64_u8(const double* x, double threshold, unsigned char* out, long n, zh_stream_t stream);

/* Retrieval (datasets/index_dataset.py:163-167): per row of scores [rows, N] (row stride ld) the k largest entries, score
 * descending, ties by ascending index — replaces torch.argsort(descending=True)[:n_images] per category.
 * idx_out int64 [rows,k]; val_out f32 [rows,k] or NULL.  k <= 1024.  The index reported for column i is
 * idx_map[row*ld + i] when idx_map != NULL (merging candidate lists that carry global image indices), else i + idx_add
 * (chunk offset); ties are broken by ascending column.  Output rows have stride out_ld >= k. */
int zh_topk_rows(const float* scores, long ld, int rows, long N, int k, const long long* idx_map, long long idx_add,
                 long long* idx_out, float* val_out, long out_ld, zh_stream_t stream);

/* Greedy per-category mask NMS on the device: networks/zutis.py:211-299 (copy: coco20k_eval.py:54-136).  inter / uni int32
 * [B,Q,Q] from zh_mask_iou_counts (IoU = inter / (uni + 1e-7) in float64 = utils/iou.py:30-32), scores f32 [B,Q], category_ids
 * int64 [B,Q].  nms_type 0 hard | 1 linear | 2 gaussian; the reference's constants are nms_threshold 0.3, sigma 0.5,
 * score_threshold 0.001.  Output per image, in the reference's emission order (category ascending, 0 skipped, then selection
 * order; empty masks skipped): out_index int32 [B,Q], out_score f64 [B,Q], out_category int64 [B,Q], out_count int32 [B]. */
int zh_mask_nms(const int* inter, const int* uni, const float* scores, const long long* category_ids, int B, int Q,
                int nms_type, double nms_threshold, double sigma, double score_threshold,
                int* out_index, double* out_score, long long* out_category, int* out_count, double* packed, const int* range_flag,
                int* zero_word, zh_stream_t stream);
/* packed (may be NULL): f64 [B, 4Q + 2] = per image [out_index (-1 past the count) | out_score | out_category | category_ids | count,
 * *range_flag] — everything the host needs in one device -> host copy.  zero_word (may be NULL): one int32 the kernel sets to 0 — the
 * cursor of the zh_mask_rle_fused_kept launched behind it (saves the caller a fill launch). */

/* Device-side run extraction for COCO RLE + boxes + areas of selected masks (masks u8 [n,H,W] row-major; sel int32
 * [n_sel] mask indices): positions int32 [n_sel, max_runs] = column-major pixel indices where the value changes;
 * nruns int32 [n_sel,2] = {#transitions, value of pixel 0}; box_area int32 [n_sel,5] = {xmin,ymin,xmax,ymax,area}.
 * Replaces the mask D2H in front of pycocotools.mask.encode / masks_to_boxes, zutis.py:288-294,446-452. */
size_t zh_mask_runs_workspace_size(int n, int W);   /* n = n_sel (zh_mask_runs) or B * Q (zh_mask_runs_kept): per-panel counts / boxes */
int zh_mask_runs(const unsigned char* masks, const int* sel, int n_sel, int H, int W, int max_runs,
                 int* positions, int* nruns, int* box_area, void* workspace, size_t workspace_bytes, zh_stream_t stream);
/* The same for the queries zh_mask_nms kept, straight from its device outputs (masks u8 [B,Q,H,W]; kept_index int32 [B,Q], kept_count
 * int32 [B]): row b*Q + j of positions / nruns / box_area describes image b's j-th kept mask; rows j >= kept_count[b] are not written.
 * packed_capacity = 0: positions int32 [B*Q, max_runs] as above.  packed_capacity > 0: positions is ONE list of packed_capacity ints —
 * the kept masks' transitions back to back (image-major, kept order; a mask contributes min(#transitions, max_runs) entries, so the
 * host finds every list from nruns alone); entries that would fall past the capacity are not written (the host sees that from the
 * same counts and asks again).  A short head of such a list travels in the same device -> host copy as the small tables. */
int zh_mask_runs_kept(const unsigned char* masks, const int* kept_index, const int* kept_count, int B, int Q, int H, int W, int max_runs,
                      int* positions, long packed_capacity, int* nruns, int* box_area, void* workspace, size_t workspace_bytes,
                      zh_stream_t stream);

/* COCO RLE strings of the kept masks on the device (= pycocotools.mask.encode(...)["counts"], zutis.py:290,448) from zh_mask_runs_kept's
 * PACKED list and nruns table: mask b*Q + j (j < kept_count[b]) gets its string at out + 5 * off + 16 * rank, off = start of its list in
 * `positions` (sum of min(transitions, max_runs) over the kept masks in front of it), rank = number of kept masks in front of it;
 * out_len int32 [B*Q] = the string's length, -1 when the mask has more than max_runs transitions or its list / string does not fit the
 * capacities (the caller encodes that mask from its pixels).  HW = pixels per mask. */
int zh_mask_rle_kept(const int* positions, long packed_capacity, const int* nruns, const int* kept_count, int B, int Q, int max_runs, long HW,
                     unsigned char* out, long out_capacity, int* out_len, zh_stream_t stream);

/* Runs + box + area + RLE string of the kept masks in ONE launch (one workgroup per kept mask, the mask held as bits in LDS): what
 * zh_mask_runs_kept + zh_mask_rle_kept produce, for masks with W <= 1024 whose bits (H*W/8 bytes), count table and run list
 * (4 * max_runs bytes) fit 150 KB of LDS (zh_mask_rle_fused_supported).  bits (may be NULL: the bytes of `masks` are packed instead) =
 * the masks bit-packed as zh_mask_iou_counts leaves them in its workspace: u64 [B*Q][(H*W + 63) / 64], bit i of word w = pixel 64 w + i
 * (row-major) is non-zero.  info int32 [B*Q, 8] = {offset of the string in `out`, its length (-1: not encoded — more than max_runs
 * transitions, or `out` is full: the caller uses zh_mask_runs for that mask), xmin, ymin, xmax, ymax, area, transitions}; rows of slots
 * j >= kept_count[b] are not written.  cursor: ONE int32 the caller has zeroed; strings are placed in order of arrival. */
int zh_mask_rle_fused_supported(int H, int W, int max_runs);
int zh_mask_rle_fused_kept(const unsigned char* masks, const unsigned long long* bits, const int* kept_index, const int* kept_count, int B, int Q,
                           int H, int W, int max_runs, unsigned char* out, long out_capacity, int* cursor, int* info, zh_stream_t stream);

/* Native launch plans (zutis_amd/plan.py): replay n recorded calls of the entry points above (op id + 24 argument words
 * each; dispatcher generated from this header) in one C loop; zh_plan_run2 alternates two plans on two streams. */
int zh_plan_run(const void* cmds, int n, zh_stream_t stream);
int zh_plan_run2(const void* cmds_a, int na, zh_stream_t stream_a, const void* cmds_b, int nb, zh_stream_t stream_b);
/* `count` plans round-robin on `count` streams (launch i of every plan before launch i+1 of any). */
int zh_plan_run_multi(const void* const* cmds, const int* n, const zh_stream_t* streams, int count);
/* Name of the entry point the library dispatches plan op `op` to (NULL if out of range): the host checks it against its
 * own table so that a stale library cannot mis-dispatch. */
const char* zh_plan_op_name(int op);

/* HOST helper: RLE string from run lengths (pycocotools rleToString). */
long zh_rle_counts_to_string_host(const long long* counts, long n, char* out, long cap);
/* HOST.  The RLE strings of n masks from zh_mask_runs' output in one call: positions int32 [n, stride] (packed = 0) or the packed
 * list of zh_mask_runs_kept (packed = 1, stride = its max_runs: row i follows row i - 1, min(transitions, stride) entries each), nruns
 * int32 [n, 2] (transitions, value of pixel 0), HW pixels per mask; strings back to back in `out` (cap bytes), offsets int64 [n + 1]; a
 * mask with more transitions than `stride` gets an empty string (the caller re-encodes it from the mask).  Returns the total length,
 * -1 when cap is too small.  Replaces pycocotools.mask.encode per kept mask (networks/zutis.py:290,448). */
long zh_rle_from_transitions_host(const int* positions, long stride, int packed, const int* nruns, long n, long HW, char* out, long cap,
                                  long long* offsets);

/* HOST helper (no GPU): COCO RLE string of one u8 [H,W] mask = pycocotools.mask.encode(np.asfortranarray(m))["counts"]
 * (zutis.py:290,448; datasets/index_dataset.py:219).  Returns the length, -1 if cap is too small. */
long zh_rle_encode_host(const unsigned char* mask, int H, int W, char* out, long cap);

#ifdef __cplusplus
}
#endif
#endif /* ZUTIS_HIP_H */
