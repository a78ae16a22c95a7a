// CLIP text tower glue (networks/clip_arch.py:534-547) and prompt ensembling (utils/extract_text_embeddings.py:98-115):
// token-embedding gather + positional add, EOT-row gather, per-category mean of unit-norm rows.  HBM-bound row kernels;
// the transformer blocks themselves run on zh_gemm_f16 / zh_attention_causal_f16 / zh_layernorm_f32.
#include "common.h"

// out[i*ctx + t][:] = table[tokens[i][t]][:] + pos[t][:]          (clip_arch.py:535-537)
__global__ __launch_bounds__(256) void embed_tokens_kernel(const long long* tokens, const float* table, const float* pos,
                                                           float* out, long rows, int ctx, int D4, int vocab) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * D4) return;
  const long r = i / D4;
  const int c = (int)(i - r * D4);
  long long tok = tokens[r];
  tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);      // ids are validated by the host wrapper; clamp keeps the read in bounds
  const f32x4 e = *(const f32x4*)(table + tok * (long)D4 * 4 + c * 4);
  const f32x4 q = *(const f32x4*)(pos + (r % ctx) * (long)D4 * 4 + c * 4);
  *(f32x4*)(out + i * 4) = e + q;
}

extern "C" int zh_embed_tokens_f32(const long long* tokens, const float* table, const float* pos, float* out, long n, int ctx,
                                   int D, int vocab, hipStream_t stream) {
  ZH_CHECK_ARG(tokens && table && pos && out && n > 0 && ctx > 0 && D > 0 && D % 4 == 0 && vocab > 0, "zh_embed_tokens_f32: bad arguments");
  const long work = n * ctx * (D / 4);
  hipLaunchKernelGGL(embed_tokens_kernel, dim3(zh_cdiv(work, 256)), dim3(256), 0, stream, tokens, table, pos, out, n * ctx, ctx, D / 4, vocab);
  ZH_CHECK_LAUNCH("zh_embed_tokens_f32");
  return ZH_OK;
}

// out[i][:] = x[i*ctx + argmax_t tokens[i][t]][:]   (first maximum, as torch.argmax; clip_arch.py:545).  One wave per row.
__global__ __launch_bounds__(64) void eot_rows_kernel(const long long* tokens, const float* x, float* out, int ctx, int D) {
  const long i = blockIdx.x;
  const int lane = threadIdx.x;
  long long best = -0x7FFFFFFFFFFFFFFFLL - 1;
  int bi = 0;
  for (int t = lane; t < ctx; t += 64) {
    const long long v = tokens[i * ctx + t];
    if (v > best) { best = v; bi = t; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const long long ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  const float* src = x + (i * ctx + bi) * (long)D;
  for (int c = lane; c < D; c += 64) out[i * (long)D + c] = src[c];
}

extern "C" int zh_eot_rows_f32(const long long* tokens, const float* x, float* out, long n, int ctx, int D, hipStream_t stream) {
  ZH_CHECK_ARG(tokens && x && out && n > 0 && ctx > 0 && D > 0, "zh_eot_rows_f32: bad arguments");
  ZH_CHECK_ARG(n < (1L << 31), "zh_eot_rows_f32: too many rows");
  hipLaunchKernelGGL(eot_rows_kernel, dim3((unsigned)n), dim3(64), 0, stream, tokens, x, out, ctx, D);
  ZH_CHECK_LAUNCH("zh_eot_rows_f32");
  return ZH_OK;
}

// out[g][:] = m / ||m||,  m = mean_t x[g][t][:]   (rows of x already unit-norm; extract_text_embeddings.py:110-112).
// One workgroup per category; the T-sum is kept in float64 (the reference's float32 cascade sum is order-dependent in the
// last bit; float64 is within 1 ulp of any order).
__global__ __launch_bounds__(256) void group_mean_l2_kernel(const float* x, float* out, int T, int E) {
  __shared__ double red[4];
  const long g = blockIdx.x;
  const float* xg = x + g * (long)T * E;
  double ss = 0.0;
  for (int c = threadIdx.x; c < E; c += 256) {
    double a = 0.0;
    for (int t = 0; t < T; ++t) a += (double)xg[(long)t * E + c];
    const float m = (float)(a / (double)T);
    out[g * (long)E + c] = m;
    ss += (double)m * (double)m;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  const float nrm = sqrtf((float)(red[0] + red[1] + red[2] + red[3]));
  for (int c = threadIdx.x; c < E; c += 256) out[g * (long)E + c] = out[g * (long)E + c] / nrm;
}

extern "C" int zh_group_mean_l2norm(const float* x, float* out, int groups, int T, int E, hipStream_t stream) {
  ZH_CHECK_ARG(x && out && groups > 0 && T > 0 && E > 0, "zh_group_mean_l2norm: bad arguments");
  hipLaunchKernelGGL(group_mean_l2_kernel, dim3(groups), dim3(256), 0, stream, x, out, T, E);
  ZH_CHECK_LAUNCH("zh_group_mean_l2norm");
  return ZH_OK;
}
