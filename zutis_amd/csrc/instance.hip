// Instance-prediction kernels (networks/zutis.py:374-420): per-query mask statistics, masked mean of the
// text-space patch tokens, and the per-query class / score.  The reference materialises
// patch_tokens[:,None] * binary_masks[...,None] = B x Q x hw x 512 fp32 (2.9 GB at B=8, zutis.py:404-406);
// here the masked mean is a streaming reduction that reads each image's tokens from L2.
#include "common.h"

// ---- per (image, query): size = #(p > thr), conf = sum(p * (p > thr)) / (size + 1e-7)   (zutis.py:390-397)
// range_flag (optional): bit 0 is set when any proposal lies outside [0, 1] (or is a NaN) — the reference's two asserts on the
// mask proposals (zutis.py:385-386), checked by the host at the predict's one synchronisation instead of with a reduction + copy of its own
// One workgroup per (image, query) row, 4 pixels per thread and pass (round 4: a wave per row walked its 4800 pixels in 75 dependent
// passes, 25 workgroups at batch 1: 26 us for 1.9 MB).  The sums are integers (counts) and a float sum whose order is fixed by the
// shape: per thread ascending pixels, then the fixed tree of block_sum4.
__device__ __forceinline__ float stats_block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void instance_stats_kernel(const float* mp, long stride_b, float thr, long rows, int Q, int M,
                                                             float* sizes, float* conf, unsigned char* binary, int* range_flag) {
  __shared__ float red[4];
  const long r = blockIdx.x;                                      // r = b*Q + q
  const int b = (int)(r / Q), q = (int)(r % Q);
  const float* p = mp + (long)b * stride_b + (long)q * M;
  unsigned char* bo = binary + r * M;
  float cnt = 0.f, s = 0.f;
  bool bad = false;
  const bool vec = (M % 4 == 0) && ((((uintptr_t)p) & 15) == 0) && ((((uintptr_t)bo) & 3) == 0);
  if (vec) {
    for (int m = threadIdx.x * 4; m < M; m += 1024) {
      const f32x4 v = *(const f32x4*)(p + m);
      unsigned pk = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bad |= !(v[e] >= 0.f && v[e] <= 1.f);
        const bool on = v[e] > thr;
        cnt += on ? 1.f : 0.f;
        s += on ? v[e] : 0.f;
        pk |= (on ? 1u : 0u) << (8 * e);
      }
      *(unsigned*)(bo + m) = pk;
    }
  } else {
    for (int m = threadIdx.x; m < M; m += 256) {
      const float v = p[m];
      bad |= !(v >= 0.f && v <= 1.f);
      const bool on = v > thr;
      cnt += on ? 1.f : 0.f;
      s += on ? v : 0.f;
      bo[m] = on ? 1 : 0;
    }
  }
  cnt = stats_block_sum(cnt, red);
  s = stats_block_sum(s, red);
  if (threadIdx.x == 0) { sizes[r] = cnt; conf[r] = s / (cnt + 1e-7f); }
  if (range_flag && __ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(range_flag, 1);
}

extern "C" int zh_instance_mask_stats(const float* mask_proposals, long stride_image, float threshold, int B, int Q, int M,
                                      float* sizes, float* confidence, unsigned char* binary, int* range_flag, hipStream_t stream) {
  ZH_CHECK_ARG(mask_proposals && sizes && confidence && binary && B > 0 && Q > 0 && M > 0, "zh_instance_mask_stats: bad arguments");
  const long rows = (long)B * Q;
  ZH_CHECK_ARG(rows < (1L << 31), "zh_instance_mask_stats: too many rows");
  hipLaunchKernelGGL(instance_stats_kernel, dim3((unsigned)rows), dim3(256), 0, stream, mask_proposals, stride_image, threshold, rows, Q, M,
                     sizes, confidence, binary, range_flag);
  ZH_CHECK_LAUNCH("zh_instance_mask_stats");
  return ZH_OK;
}

// ---- avg[b,q,:] = sum_m binary[b,q,m] * tokens[b,m,:] / (size[b,q] + 1e-7)   (zutis.py:404-406)
// block = (tile of QT queries, image, chunk of MCH pixels); thread owns CPT channels; the tile's mask bytes of the chunk are
// staged in LDS.  Per-chunk partial sums go to the workspace [chunks][B*Q][E] and masked_mean_reduce_kernel adds them in chunk
// order (deterministic, independent of the batch) and divides.  (Round 3: the first version walked ALL pixels in one block per
// (query tile, image) — 10 blocks at batch 1, the COCO-20K evaluation's regime, 4800 dependent iterations: 1.8 ms of the
// 3.5 ms instance predict.)
#define QT 10      // (round 4: 20 queries per block with 16 rows of loads in flight: 31 -> 60 us — 190 blocks, one round, each twice as long)
#define MCH 64     // pixels per block (round 4: 128 -> 64: twice the workgroups, half the dependent load rounds per workgroup)
template <int CPT>
__global__ __launch_bounds__(256) void masked_mean_kernel(const float* tokens, const unsigned char* binary, float* partial, int Q, int M, int E,
                                                          long rows) {
  __shared__ unsigned char sm[QT][MCH];
  const int b = blockIdx.y, q0 = blockIdx.x * QT, m0 = blockIdx.z * MCH;
  const int nq = min(QT, Q - q0), mc = min(MCH, M - m0);
  const float* tk = tokens + (long)b * M * E;
  float acc[QT][CPT];
#pragma unroll
  for (int q = 0; q < QT; ++q)
#pragma unroll
    for (int c = 0; c < CPT; ++c) acc[q][c] = 0.f;
  for (int i = threadIdx.x; i < QT * MCH; i += 256) {
    const int q = i / MCH, m = i - q * MCH;
    sm[q][m] = (q < nq && m < mc) ? binary[((long)b * Q + q0 + q) * M + m0 + m] : 0;
  }
  __syncthreads();
#pragma unroll 8                                         // eight rows of loads in flight (same summation order per accumulator; 16: 208 VGPRs)
  for (int m = 0; m < mc; ++m) {
    float v[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const int ch = threadIdx.x + 256 * c;
      v[c] = ch < E ? tk[(long)(m0 + m) * E + ch] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < QT; ++q) {
      const float f = sm[q][m] ? 1.f : 0.f;
#pragma unroll
      for (int c = 0; c < CPT; ++c) acc[q][c] += f * v[c];
    }
  }
  float* po = partial + ((long)blockIdx.z * rows + (long)b * Q + q0) * E;
  for (int q = 0; q < nq; ++q)
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const int ch = threadIdx.x + 256 * c;
      if (ch < E) po[(long)q * E + ch] = acc[q][c];
    }
}
__global__ __launch_bounds__(256) void masked_mean_reduce_kernel(const float* partial, const float* sizes, float* avg, long rows, int E, int chunks) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * E) return;
  float s = 0.f;
#pragma unroll 8
  for (int z = 0; z < chunks; ++z) s += partial[(long)z * rows * E + i];
  avg[i] = s * (1.0f / (sizes[i / E] + 1e-7f));
}

extern "C" size_t zh_masked_mean_workspace_size(int B, int Q, int M, int E) {
  return (size_t)zh_cdiv(M, MCH) * B * Q * E * sizeof(float);
}
extern "C" int zh_masked_mean_tokens(const float* tokens, const unsigned char* binary, const float* sizes, float* avg,
                                     int B, int Q, int M, int E, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  ZH_CHECK_ARG(tokens && binary && sizes && avg && B > 0 && Q > 0 && M > 0 && E > 0, "zh_masked_mean_tokens: bad arguments");
  ZH_CHECK_ARG(E <= 1024 && B < 65536, "zh_masked_mean_tokens: E=%d > 1024 unsupported", E);
  const int chunks = zh_cdiv(M, MCH);
  ZH_CHECK_ARG(chunks < 65536, "zh_masked_mean_tokens: M=%d too large", M);
  if (!workspace || workspace_bytes < zh_masked_mean_workspace_size(B, Q, M, E)) {
    zh_set_error("zh_masked_mean_tokens: workspace too small (%zu < %zu)", workspace_bytes, zh_masked_mean_workspace_size(B, Q, M, E));
    return ZH_ERR_WORKSPACE;
  }
  float* partial = (float*)workspace;
  const long rows = (long)B * Q;
  dim3 grid(zh_cdiv(Q, QT), B, chunks);
  if (E <= 256) hipLaunchKernelGGL(masked_mean_kernel<1>, grid, dim3(256), 0, stream, tokens, binary, partial, Q, M, E, rows);
  else if (E <= 512) hipLaunchKernelGGL(masked_mean_kernel<2>, grid, dim3(256), 0, stream, tokens, binary, partial, Q, M, E, rows);
  else hipLaunchKernelGGL(masked_mean_kernel<4>, grid, dim3(256), 0, stream, tokens, binary, partial, Q, M, E, rows);
  hipLaunchKernelGGL(masked_mean_reduce_kernel, dim3(zh_cdiv(rows * E, 256)), dim3(256), 0, stream, partial, sizes, avg, rows, E, chunks);
  ZH_CHECK_LAUNCH("zh_masked_mean_tokens");
  return ZH_OK;
}

// ---- per (image, query): v = avg / (||avg|| + 1e-7); prob_n = sigmoid(T * text_n . v);
//      category = argmax_n (first max), score = conf * max_n prob      (zutis.py:409-420)
__global__ __launch_bounds__(256) void instance_classify_kernel(const float* avg, const float* text, const float* conf, float temperature,
                                                                int n, int E, long long* category, float* score) {
  __shared__ float sv[1024];
  __shared__ float red[4];
  __shared__ float bestv[4];
  __shared__ int besti[4];
  const long r = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* a = avg + r * E;
  float q = 0.f;
  for (int c = threadIdx.x; c < E; c += 256) { const float v = a[c]; sv[c] = v; q += v * v; }
  q = wave_sum(q);
  if (lane == 0) red[wave] = q;
  __syncthreads();
  const float inv = 1.0f / (sqrtf((red[0] + red[1]) + (red[2] + red[3])) + 1e-7f);
  float bv = -1.f;
  int bi = 0x7fffffff;
  // CPP classes per pass (cls, cls + 4, ...): their rows are loaded together — one class at a time was 21 dependent passes of
  // load -> reduce -> exp per wave, 48 us for 100 queries x 81 classes at batch 1.  Each class's dot product keeps its summation
  // order and the classes are compared in ascending order, as before: bit-identical categories and scores.
  constexpr int CPP = 6;                                  // classes per pass and wave (round 4: 3 -> 6, 81 classes in 4 passes instead of 7)
  for (int cls = wave; cls < n; cls += 4 * CPP) {
    const float* t[CPP];
    float d[CPP];
#pragma unroll
    for (int j = 0; j < CPP; ++j) { t[j] = text + (long)(cls + 4 * j < n ? cls + 4 * j : cls) * E; d[j] = 0.f; }
    for (int c = lane; c < E; c += 64) {                     // (unrolled by 8 — 48 loads in flight — this kernel ran 40 us instead of 22)
      const float x = sv[c] * inv;
#pragma unroll
      for (int j = 0; j < CPP; ++j) d[j] += t[j][c] * x;
    }
#pragma unroll
    for (int j = 0; j < CPP; ++j) {
      const float dj = wave_sum(d[j]);
      const float pr = 1.0f / (1.0f + expf(-temperature * dj));
      if (cls + 4 * j < n && pr > bv) { bv = pr; bi = cls + 4 * j; }          // classes ascend within a wave -> first max kept
    }
  }
  if (lane == 0) { bestv[wave] = bv; besti[wave] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float v = bestv[0];
    int i = besti[0];
    for (int w = 1; w < 4; ++w)
      if (bestv[w] > v || (bestv[w] == v && besti[w] < i)) { v = bestv[w]; i = besti[w]; }
    category[r] = i;
    score[r] = conf[r] * v;
  }
}

extern "C" int zh_instance_classify(const float* avg, const float* text, const float* confidence, float temperature,
                                    int rows, int n_classes, int E, long long* category, float* score, hipStream_t stream) {
  ZH_CHECK_ARG(avg && text && confidence && category && score && rows > 0 && n_classes > 0 && E > 0 && E <= 1024,
               "zh_instance_classify: bad arguments");
  hipLaunchKernelGGL(instance_classify_kernel, dim3(rows), dim3(256), 0, stream, avg, text, confidence, temperature, n_classes, E,
                     category, score);
  ZH_CHECK_LAUNCH("zh_instance_classify");
  return ZH_OK;
}

// ---- pairwise mask IoU on bit-packed masks: iou[i][j] = |a_i & a_j| / (|a_i | a_j| + 1e-7)  (utils/iou.py:6-37)
//      masks u8 {0,1} [n, P] -> inter/union counts int32 [n, n] (exact integers; the float divide happens on the host
//      in float64 exactly as numpy does).
__global__ __launch_bounds__(256) void mask_pack_kernel(const unsigned char* masks, unsigned long long* packed, long P, long W64, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;   // one thread per 64-bit word
  if (idx >= total) return;
  const long n = idx / W64, w = idx - n * W64;
  const unsigned char* m = masks + n * P + w * 64;
  unsigned long long bits = 0;
  const long lim = P - w * 64 < 64 ? P - w * 64 : 64;
  if (lim == 64 && (((uintptr_t)m) & 15) == 0) {
    // four 16-byte loads instead of 64 byte loads (66 us for 100 masks of 480 x 640 at batch 1): a word of four {0,1} bytes
    // b0..b3 times 0x01020408 has b0 | b1 << 1 | b2 << 2 | b3 << 3 in bits 24..27 (the other partial products land on
    // distinct lower bits: no carries)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint4 v = *(const uint4*)(m + q * 16);
      const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned t = ((w4[k] & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w4[k];
        const unsigned nz = (t >> 7) & 0x01010101u;
        bits |= (unsigned long long)(((nz * 0x01020408u) >> 24) & 0xFu) << (q * 16 + k * 4);
      }
    }
  } else {
    for (long i = 0; i < lim; ++i) bits |= (unsigned long long)(m[i] != 0) << i;
  }
  packed[idx] = bits;
}

// A block = a 4 x 4 tile of mask pairs (i0 .. i0+3) x (j0 .. j0+3), tiles on or above the diagonal: a pass loads eight 64-bit words
// per thread for sixteen (intersection, union) popcounts (round 4: one block per pair loaded two words per popcount pair, 390 MB of L2
// reads for 100 masks of 480 x 640: 30 us).  Integer sums: any order gives the same counts.
#define IOU_T 4
__global__ __launch_bounds__(256) void mask_iou_counts_kernel(const unsigned long long* packed, int n, long W64, int* inter, int* uni) {
  const int ti = blockIdx.x, tj = blockIdx.y;
  if (tj < ti) return;
  const int i0 = ti * IOU_T, j0 = tj * IOU_T;
  int ci[IOU_T][IOU_T], cu[IOU_T][IOU_T];
#pragma unroll
  for (int a = 0; a < IOU_T; ++a)
#pragma unroll
    for (int b = 0; b < IOU_T; ++b) { ci[a][b] = 0; cu[a][b] = 0; }
#pragma unroll 4                                             // 32 loads in flight per thread (19 dependent rounds for 480 x 640 otherwise)
  for (long w = threadIdx.x; w < W64; w += 256) {
    unsigned long long x[IOU_T], y[IOU_T];
#pragma unroll
    for (int a = 0; a < IOU_T; ++a) {
      x[a] = packed[(long)min(i0 + a, n - 1) * W64 + w];
      y[a] = packed[(long)min(j0 + a, n - 1) * W64 + w];
    }
#pragma unroll
    for (int a = 0; a < IOU_T; ++a)
#pragma unroll
      for (int b = 0; b < IOU_T; ++b) { ci[a][b] += __popcll(x[a] & y[b]); cu[a][b] += __popcll(x[a] | y[b]); }
  }
  __shared__ int red[4][2 * IOU_T * IOU_T];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int a = 0; a < IOU_T; ++a)
#pragma unroll
    for (int b = 0; b < IOU_T; ++b) {
      int v = ci[a][b], u = cu[a][b];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { v += __shfl_xor(v, o, 64); u += __shfl_xor(u, o, 64); }
      if (lane == 0) { red[wave][2 * (a * IOU_T + b)] = v; red[wave][2 * (a * IOU_T + b) + 1] = u; }
    }
  __syncthreads();
  if (threadIdx.x < IOU_T * IOU_T) {
    const int a = threadIdx.x / IOU_T, b = threadIdx.x % IOU_T;
    const int i = i0 + a, j = j0 + b;
    if (i < n && j < n) {
      const int k = 2 * (a * IOU_T + b);
      const int I = red[0][k] + red[1][k] + red[2][k] + red[3][k], U = red[0][k + 1] + red[1][k + 1] + red[2][k + 1] + red[3][k + 1];
      inter[(long)i * n + j] = I; inter[(long)j * n + i] = I;
      uni[(long)i * n + j] = U; uni[(long)j * n + i] = U;
    }
  }
}

extern "C" size_t zh_mask_iou_workspace_size(int n, long pixels) { return (size_t)n * ((pixels + 63) / 64) * 8; }

extern "C" int zh_mask_iou_counts(const unsigned char* masks, int n, long pixels, int* inter, int* uni,
                                  void* workspace, size_t workspace_bytes, hipStream_t stream) {
  ZH_CHECK_ARG(masks && inter && uni && n > 0 && pixels > 0 && n < 65536, "zh_mask_iou_counts: bad arguments");
  const long W64 = (pixels + 63) / 64;
  if (!workspace || workspace_bytes < zh_mask_iou_workspace_size(n, pixels)) {
    zh_set_error("zh_mask_iou_counts: workspace too small");
    return ZH_ERR_WORKSPACE;
  }
  hipLaunchKernelGGL(mask_pack_kernel, dim3(zh_cdiv((long)n * W64, 256)), dim3(256), 0, stream, masks, (unsigned long long*)workspace, pixels, W64, (long)n * W64);
  const int nt = zh_cdiv(n, IOU_T);
  hipLaunchKernelGGL(mask_iou_counts_kernel, dim3(nt, nt), dim3(256), 0, stream, (const unsigned long long*)workspace, n, W64, inter, uni);
  ZH_CHECK_LAUNCH("zh_mask_iou_counts");
  return ZH_OK;
}

// ---- run-length transitions + box + area of selected masks ON THE DEVICE (replaces the B x Q x H x W mask D2H in front of
//      pycocotools.mask.encode / masks_to_boxes, networks/zutis.py:288-294,446-452).  COCO RLE runs are column-major:
//      the mask is cut into 64-column panels staged through LDS (coalesced row reads); the value changes per column segment are
//      counted, prefix-summed, and the column-major pixel positions where the value changes are written.
//      counts = diff([0, positions..., H*W]) with a leading 0-run inserted when pixel 0 is set (host, tiny).
// One block per (mask, 64-column panel), TWO launches (round 4).  Round 3's single launch had every panel's block count the transitions of
// ALL columns left of its panel from global memory (the last panel re-read 90 % of the mask) and the first panel's block scan the whole
// mask once more for the box and the area: 55 us for 17 kept masks of 480 x 640, the critical path being those two long blocks.  Now
//   launch 1 (count): a block stages its panel [H][64] in LDS (coalesced row reads), four threads per column count the transitions of
//                     their quarter of the rows (the H-1 -> 0 wrap to the previous column included), and the panel's own pixels give its
//                     part of the box / area: per-panel results into a small table;
//   launch 2 (emit):  a block sums the counts of the panels to its left (<= W / 64 integers), stages its panel again, a 256-entry scan
//                     in (column, segment) order places its transitions and a second walk writes the column-major positions; the
//                     last panel's block writes the total, the first one folds the per-panel boxes / areas.
#define RUNS_PANEL 64
#define RUNS_SEG 4
__device__ __forceinline__ unsigned zh_nz_bytes(unsigned w) {   // 0x01 in every byte of w that is non-zero
  const unsigned t = ((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w;
  return (t >> 7) & 0x01010101u;
}
// count != NULL (zh_mask_runs_kept): entry mi = b * Q + j is the j-th kept query of image b — sel[mi] is its query index (zh_mask_nms'
// out_index), entries j >= count[b] do nothing: the NMS result never visits the host before the runs are extracted.
struct RunsArgs {
  const unsigned char* masks; const int* sel; const int* count; int Q, H, W, max_runs, npanel;
  int* positions; int* nruns; int* box_area;
  int* pcount;       // [n][npanel] transitions per panel
  int* pbox;         // [n][npanel][5] xmin, ymin, xmax, ymax, area of the panel's pixels
  long packed_cap;   // > 0 (kept form only): `positions` is ONE list of that many ints, the kept masks' transitions back to back
};
// stages the panel (values normalised to {0,1}) and returns, per thread = (column c = tid / 4, row quarter sg = tid % 4), the number of
// transitions of its segment and the value in front of it
__device__ __forceinline__ int runs_stage_and_count(const RunsArgs& a, const unsigned char* m, unsigned char* sm, int x0, int pw, int& y0, int& y1,
                                                    unsigned char& prev0) {
  const int tid = threadIdx.x, H = a.H, W = a.W;
  const bool vec = (W % 16 == 0) && (((uintptr_t)m & 15) == 0);
  if (vec && pw % 16 == 0) {
    constexpr int cpr = RUNS_PANEL / 16;
#pragma unroll 8
    for (int i = tid; i < H * cpr; i += 256) {
      const int y = i / cpr, c16 = i - y * cpr;
      uint4 v = {0u, 0u, 0u, 0u};
      if (c16 * 16 < pw) v = *(const uint4*)(m + (long)y * W + x0 + c16 * 16);
      uint4 o = {zh_nz_bytes(v.x), zh_nz_bytes(v.y), zh_nz_bytes(v.z), zh_nz_bytes(v.w)};
      *(uint4*)(sm + (long)y * RUNS_PANEL + c16 * 16) = o;
    }
  } else {
    for (int i = tid; i < H * RUNS_PANEL; i += 256) {
      const int y = i / RUNS_PANEL, c = i - y * RUNS_PANEL;
      sm[i] = c < pw ? (m[(long)y * W + x0 + c] != 0) : 0;
    }
  }
  __syncthreads();
  const int c = tid / RUNS_SEG, sg = tid % RUNS_SEG;
  const int hs = (H + RUNS_SEG - 1) / RUNS_SEG;
  y0 = min(H, sg * hs); y1 = min(H, y0 + hs);
  prev0 = 0;
  int mine = 0;
  if (c < pw && y0 < y1) {
    if (y0 > 0) prev0 = sm[(y0 - 1) * RUNS_PANEL + c];
    else if (c > 0) prev0 = sm[(H - 1) * RUNS_PANEL + c - 1];
    else prev0 = x0 > 0 ? (unsigned char)(m[(long)(H - 1) * W + x0 - 1] != 0) : sm[0];   // pixel 0 starts the list: no transition
    unsigned char prev = prev0;
#pragma unroll 8
    for (int y = y0; y < y1; ++y) {
      const unsigned char v = sm[y * RUNS_PANEL + c];
      mine += v != prev;
      prev = v;
    }
  }
  return mine;
}
__device__ __forceinline__ const unsigned char* runs_mask(const RunsArgs& a, int mi) {
  return a.masks + (a.count ? (long)(mi / a.Q) * a.Q + a.sel[mi] : (long)a.sel[mi]) * a.H * a.W;
}
__global__ __launch_bounds__(256) void mask_runs_count_kernel(RunsArgs a) {
  extern __shared__ unsigned char sm[];                    // [H][64] panel
  __shared__ int s_cnt, s_minx, s_maxx, s_miny, s_maxy, s_area;
  const int tid = threadIdx.x, mi = blockIdx.x, pnl = blockIdx.y;
  if (a.count && mi % a.Q >= a.count[mi / a.Q]) return;   // whole workgroup, before any barrier
  const unsigned char* m = runs_mask(a, mi);
  const int x0 = pnl * RUNS_PANEL, pw = min(RUNS_PANEL, a.W - x0);
  if (tid == 0) { s_cnt = 0; s_minx = a.W; s_maxx = -1; s_miny = a.H; s_maxy = -1; s_area = 0; }
  int y0, y1; unsigned char prev0;
  const int mine = runs_stage_and_count(a, m, sm, x0, pw, y0, y1, prev0);   // (contains the barrier that publishes the initial values)
  // box / area of the panel's own pixels: thread = (column, row quarter)
  const int c = tid / RUNS_SEG;
  int area = 0, miny = a.H, maxy = -1;
  if (c < pw)
    for (int y = y0; y < y1; ++y)
      if (sm[y * RUNS_PANEL + c]) { ++area; miny = min(miny, y); maxy = max(maxy, y); }
  if (mine) atomicAdd(&s_cnt, mine);
  if (area) { atomicAdd(&s_area, area); atomicMin(&s_minx, x0 + c); atomicMax(&s_maxx, x0 + c); atomicMin(&s_miny, miny); atomicMax(&s_maxy, maxy); }
  __syncthreads();
  if (tid == 0) {
    a.pcount[(long)mi * a.npanel + pnl] = s_cnt;
    int* b = a.pbox + ((long)mi * a.npanel + pnl) * 5;
    b[0] = s_minx; b[1] = s_miny; b[2] = s_maxx; b[3] = s_maxy; b[4] = s_area;
  }
}
__global__ __launch_bounds__(256) void mask_runs_emit_kernel(RunsArgs a) {
  extern __shared__ unsigned char sm[];
  __shared__ int s_red[256];
  const int tid = threadIdx.x, mi = blockIdx.x, pnl = blockIdx.y;
  if (a.count && mi % a.Q >= a.count[mi / a.Q]) return;
  const unsigned char* m = runs_mask(a, mi);
  int* pos = a.positions + (long)mi * a.max_runs;
  long room = a.max_runs;                                  // entries this mask may write
  if (a.packed_cap > 0) {
    // packed form: mask mi's list starts where the lists of the kept masks in front of it (image-major, kept order) end; each of those
    // is min(its transitions, max_runs) long — summed here from the count kernel's per-panel counts (<= B*Q*npanel ints, L2-resident)
    long part = 0;
    for (int mp = tid; mp < mi; mp += 256) {
      if (mp % a.Q >= a.count[mp / a.Q]) continue;
      int t = 0;
      for (int p = 0; p < a.npanel; ++p) t += a.pcount[(long)mp * a.npanel + p];
      part += min(t, a.max_runs);
    }
    s_red[tid] = (int)part;                                // < 2^31: B*Q*max_runs is checked by the host entry
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
      if (tid < st) s_red[tid] += s_red[tid + st];
      __syncthreads();
    }
    const long off = s_red[0];
    __syncthreads();
    pos = a.positions + off;
    room = min((long)a.max_runs, a.packed_cap - off);      // <= 0: the list does not fit any more (the host sees it from the counts)
  }
  const int x0 = pnl * RUNS_PANEL, pw = min(RUNS_PANEL, a.W - x0);
  int base = 0;
  for (int p = 0; p < pnl; ++p) base += a.pcount[(long)mi * a.npanel + p];
  int y0, y1; unsigned char prev0;
  const int mine = runs_stage_and_count(a, m, sm, x0, pw, y0, y1, prev0);
  s_red[tid] = mine;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {                      // inclusive scan (Hillis - Steele), entry e = column * RUNS_SEG + segment == tid
    const int t = tid >= d ? s_red[tid - d] : 0;
    __syncthreads();
    s_red[tid] += t;
    __syncthreads();
  }
  const int total = base + s_red[255];
  if (mine) {
    const int c = tid / RUNS_SEG;
    int o = base + s_red[tid] - mine;
    unsigned char prev = prev0;
#pragma unroll 8
    for (int y = y0; y < y1; ++y) {
      const unsigned char v = sm[y * RUNS_PANEL + c];
      if (v != prev) { if (o < room) pos[o] = (x0 + c) * a.H + y; ++o; }
      prev = v;
    }
  }
  if (pnl == a.npanel - 1 && tid == 0) {
    a.nruns[mi * 2] = total;                               // number of transitions (may exceed max_runs => host fallback)
    a.nruns[mi * 2 + 1] = m[0] != 0;                       // value of pixel 0
  }
  if (pnl == 0 && tid == 0) {                              // fold the per-panel boxes / areas
    int minx = a.W, miny = a.H, maxx = -1, maxy = -1, area = 0;
    for (int p = 0; p < a.npanel; ++p) {
      const int* b = a.pbox + ((long)mi * a.npanel + p) * 5;
      minx = min(minx, b[0]); miny = min(miny, b[1]); maxx = max(maxx, b[2]); maxy = max(maxy, b[3]); area += b[4];
    }
    int* o = a.box_area + mi * 5;
    o[0] = minx; o[1] = miny; o[2] = maxx; o[3] = maxy; o[4] = area;
  }
}

static int launch_mask_runs(const char* what, RunsArgs a, int n, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  const size_t lds = ((size_t)a.H * RUNS_PANEL + 15) & ~(size_t)15;
  ZH_CHECK_ARG(lds <= 150 * 1024, "%s: H=%d too tall for the LDS panel", what, a.H);
  a.npanel = zh_cdiv(a.W, RUNS_PANEL);
  ZH_CHECK_ARG(a.npanel <= 65535, "%s: mask too wide", what);
  const size_t need = (size_t)n * a.npanel * 6 * sizeof(int);
  if (!workspace || workspace_bytes < need) {
    zh_set_error("%s: workspace too small (%zu < %zu)", what, workspace_bytes, need);
    return ZH_ERR_WORKSPACE;
  }
  a.pcount = (int*)workspace;
  a.pbox = a.pcount + (size_t)n * a.npanel;
  if (lds > 64 * 1024) {
    (void)hipFuncSetAttribute((const void*)mask_runs_count_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)mask_runs_emit_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  hipLaunchKernelGGL(mask_runs_count_kernel, dim3(n, a.npanel), dim3(256), lds, stream, a);
  hipLaunchKernelGGL(mask_runs_emit_kernel, dim3(n, a.npanel), dim3(256), lds, stream, a);
  ZH_CHECK_LAUNCH(what);
  return ZH_OK;
}
extern "C" size_t zh_mask_runs_workspace_size(int n, int W) { return (size_t)n * zh_cdiv(W, RUNS_PANEL) * 6 * sizeof(int); }

extern "C" int zh_mask_runs(const unsigned char* masks, const int* sel, int n_sel, int H, int W, int max_runs,
                            int* positions, int* nruns, int* box_area, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  ZH_CHECK_ARG(masks && sel && positions && nruns && box_area && n_sel > 0 && H > 0 && W > 0 && max_runs > 0, "zh_mask_runs: bad arguments");
  ZH_CHECK_ARG((long)H * W < (1L << 31), "zh_mask_runs: mask too large");
  RunsArgs a{masks, sel, nullptr, 1, H, W, max_runs, 0, positions, nruns, box_area, nullptr, nullptr, 0};
  return launch_mask_runs("zh_mask_runs", a, n_sel, workspace, workspace_bytes, stream);
}

extern "C" int zh_mask_runs_kept(const unsigned char* masks, const int* kept_index, const int* kept_count, int B, int Q, int H, int W, int max_runs,
                                 int* positions, long packed_capacity, int* nruns, int* box_area, void* workspace, size_t workspace_bytes,
                                 hipStream_t stream) {
  ZH_CHECK_ARG(masks && kept_index && kept_count && positions && nruns && box_area && B > 0 && Q > 0 && H > 0 && W > 0 && max_runs > 0,
               "zh_mask_runs_kept: bad arguments");
  ZH_CHECK_ARG((long)H * W < (1L << 31) && (long)B * Q < (1L << 31), "zh_mask_runs_kept: mask / batch too large");
  ZH_CHECK_ARG(packed_capacity >= 0 && (packed_capacity == 0 || (long)B * Q * max_runs < (1L << 31)),
               "zh_mask_runs_kept: packed_capacity=%ld (needs B*Q*max_runs < 2^31)", packed_capacity);
  RunsArgs a{masks, kept_index, kept_count, Q, H, W, max_runs, 0, positions, nruns, box_area, nullptr, nullptr, packed_capacity};
  return launch_mask_runs("zh_mask_runs_kept", a, B * Q, workspace, workspace_bytes, stream);
}

// ---- COCO RLE strings of the kept masks, on the device (pycocotools rleEncode + rleToString, the strings of zutis.py:290,448).
// Input: zh_mask_runs_kept's packed list (column-major pixel indices where the value changes) and its nruns table.  One workgroup per
// kept mask: run k = edge(k + 1) - edge(k) with edge(0) = 0, edge(i) = positions[i - 1], edge(nt + 1) = H*W (plus a leading empty run of
// zeros when pixel 0 is set: the format starts with zeros); x_k = run_k - run_{k-2} for k > 2, else run_k; x is written as little-endian
// 5-bit groups, bit 5 = "more", + 48 — the number of groups is a function of x alone, so a block scan of the group counts places every
// run's characters.  Mask mi's string starts at 5 * off(mi) + 16 * rank(mi) in `out` (off = the start of its list in the packed
// positions, rank = kept masks in front of it: both follow from nruns and the counts, on the host too), out_len[mi] = its length, or
// -1 when the mask has more than max_runs transitions or its list / string lies past a capacity (the host encodes that mask itself).
struct RleArgs {
  const int* positions; long packed_cap; const int* nruns; const int* count; int Q, max_runs; long HW;
  unsigned char* out; long out_cap; int* out_len;
};
#define RLE_LDS_RUNS 8192
__device__ __forceinline__ int rle_groups(long x) {
  int n = 0;
  bool more = true;
  while (more) {
    const long ch = x & 0x1f;
    x >>= 5;
    more = (ch & 0x10) ? x != -1 : x != 0;
    ++n;
  }
  return n;
}
__global__ __launch_bounds__(256) void mask_rle_kernel(RleArgs a) {
  __shared__ int s_a[256], s_b[256];
  __shared__ int s_w[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, mi = blockIdx.x;
  if (mi % a.Q >= a.count[mi / a.Q]) return;               // whole workgroup, before any barrier
  int part = 0, rank = 0;
  for (int mp = tid; mp < mi; mp += 256)
    if (mp % a.Q < a.count[mp / a.Q]) { part += min(a.nruns[2 * mp], a.max_runs); ++rank; }
  s_a[tid] = part; s_b[tid] = rank;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) { s_a[tid] += s_a[tid + st]; s_b[tid] += s_b[tid + st]; }
    __syncthreads();
  }
  const long off = s_a[0], cstart = 5L * s_a[0] + 16L * s_b[0];
  const int nt = a.nruns[2 * mi], lead = a.nruns[2 * mi + 1] != 0 ? 1 : 0;
  const int nc = nt + 1 + lead;
  if (nt > a.max_runs || off + nt > a.packed_cap || cstart + 5L * nc > a.out_cap) {
    if (tid == 0) a.out_len[mi] = -1;
    return;
  }
  // the mask's list goes to LDS in one sweep of wide loads (every run is read four times below: 25 -> ~10 us for the fixture's
  // 1750-transition masks); lists longer than the LDS copy are read where they lie
  __shared__ int s_p[RLE_LDS_RUNS];
  const int* p = a.positions + off;
  if (nt <= RLE_LDS_RUNS) {
    for (int i = tid; i < nt; i += 256) s_p[i] = p[i];
    __syncthreads();
    p = s_p;
  }
  unsigned char* o = a.out + cstart;
  auto run = [&](int k) -> long {                           // k in [0, nc)
    if (lead) { if (k == 0) return 0; --k; }
    const long lo = k == 0 ? 0 : p[k - 1], hi = k == nt ? a.HW : p[k];
    return hi - lo;
  };
  int base = 0;
  for (int k0 = 0; k0 < nc; k0 += 256) {
    const int k = k0 + tid;
    long x = 0;
    int n = 0;
    if (k < nc) { x = run(k); if (k > 2) x -= run(k - 2); n = rle_groups(x); }
    int inc = n;                                            // inclusive scan: within the wave by shuffles, across the four waves through LDS
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
    __syncthreads();                                        // (the previous chunk's s_w reads are done)
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    int wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += s_w[w];
    const int total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    if (k < nc) {
      unsigned char* q = o + base + wbase + inc - n;
      bool more = true;
      while (more) {
        long ch = x & 0x1f;
        x >>= 5;
        more = (ch & 0x10) ? x != -1 : x != 0;
        if (more) ch |= 0x20;
        *q++ = (unsigned char)(ch + 48);
      }
    }
    base += total;
  }
  if (tid == 0) a.out_len[mi] = base;
}
extern "C" int zh_mask_rle_kept(const int* positions, long packed_capacity, const int* nruns, const int* kept_count, int B, int Q, int max_runs,
                                long HW, unsigned char* out, long out_capacity, int* out_len, hipStream_t stream) {
  ZH_CHECK_ARG(positions && nruns && kept_count && out && out_len && B > 0 && Q > 0 && max_runs > 0 && HW > 0 && packed_capacity > 0 &&
               out_capacity > 0, "zh_mask_rle_kept: bad arguments");
  ZH_CHECK_ARG((long)B * Q * max_runs < (1L << 31) / 5 && HW < (1L << 31), "zh_mask_rle_kept: B*Q*max_runs or H*W too large");
  RleArgs a{positions, packed_capacity, nruns, kept_count, Q, max_runs, HW, out, out_capacity, out_len};
  hipLaunchKernelGGL(mask_rle_kernel, dim3(B * Q), dim3(256), 0, stream, a);
  ZH_CHECK_LAUNCH("zh_mask_rle_kept");
  return ZH_OK;
}

// ---- the same three steps (runs, box / area, RLE string) of a kept mask in ONE workgroup and ONE launch (round 4; the batch-1 predict's tail was
// count 27 + emit 18 + string 21 us, each a chain of a dozen dependent ~1-us steps on 17 busy workgroups).  With one workgroup per mask the
// per-pixel walk of the panel kernels would be 17 CUs doing what 170 did, so everything here is word-parallel on the mask's BITS (row-major,
// H*W/8 bytes in LDS: 38 KB for 480x640; copied from the IoU step's bit-packed workspace when the caller has it, else packed from the bytes
// with 16-byte loads).  A transition of the column-major order is a bit of T = B xor (B moved down one row) (row 0 against the last row of
// the column to the left).  A thread owns (32-column panel p, 16-row chunk s):
//   pass 1: its 16 words of T go through a bit-sliced vertical counter (5 planes, registers): per-column counts of the chunk -> part[s][c];
//   then a per-column prefix over the chunks (in place) and a block scan over the columns give every (column, chunk)'s place in the list;
//   pass 2: the 16 words again (registers); per column that has a transition in the chunk, its rows in ascending order -> list[place++];
//   string: mask_rle_kernel's steps from the LDS list into an LDS buffer (one pass: the length falls out of the scan), an atomic cursor
//           places it in `out` (order of arrival: the host reads every string's offset from `info`), a coalesced copy writes it.
// No LDS read-modify-write anywhere.  info int32 [B*Q][8] = {string offset, string length (-1: not encoded — more than max_runs
// transitions or `out` full), xmin, ymin, xmax, ymax, area, transitions}; rows of slots past an image's count are not written.
struct FusedRleArgs {
  const unsigned char* masks; const unsigned long long* bits; const int* sel; const int* count; int Q, H, W, max_runs;
  unsigned char* out; long out_cap; int* cursor; int* info;
};
#define FRLE_ROWS 16                                         // rows per chunk (= counts per (chunk, column) fit the 5-plane counter)
#ifdef ZH_FRLE_STAMP   // developer build (tools/rle_fused_stamp.py): thread 0's 100-MHz clock at the phase boundaries, in the LAST 64 B x slot of `out`
#define FRLE_STAMP(i) do { if (tid == 0) ((long long*)(a.out + a.out_cap - 64L * (mi + 1)))[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FRLE_STAMP(i) do { } while (0)
#endif
__device__ __forceinline__ unsigned rle_bits4(unsigned w) {  // bit k = byte k of w is non-zero
  const unsigned nz = zh_nz_bytes(w);
  return (nz & 1u) | ((nz >> 7) & 2u) | ((nz >> 14) & 4u) | ((nz >> 21) & 8u);
}
struct FrleLayout { long n64; int P, S, Wp; size_t o_part, o_coloff, o_list, total; };
__host__ __device__ static inline FrleLayout frle_layout(int H, int W, int max_runs) {
  FrleLayout L;
  L.n64 = ((long)H * W + 63) >> 6;
  L.P = (W + 31) >> 5;                                       // 32-column panels
  L.Wp = ((W + 63) >> 6) * 64; L.S = (H + FRLE_ROWS - 1) / FRLE_ROWS;
  L.o_part = (size_t)(L.n64 + 2) * 8;                        // u64 bits[n64 + 2] (two zero words behind the last: funnel reads)
  size_t r0 = L.o_part + (size_t)L.S * L.Wp * 2;             // u16 part[S][Wp]
  const size_t cb = ((size_t)5 * (max_runs + 2) + 15) & ~(size_t)15;
  if (r0 < cb) r0 = cb;                                      // the string is built over bits + part once they are dead
  L.o_coloff = r0;
  L.o_list = L.o_coloff + (size_t)L.Wp * 4;                  // int coloff[Wp]
  L.total = L.o_list + (size_t)max_runs * 4;                 // int list[max_runs]
  return L;
}
__global__ __launch_bounds__(1024) void mask_rle_fused_kernel(FusedRleArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
  __shared__ int s_w[16], s_red[5][16];
  __shared__ int s_off;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x, nw = nthr >> 6, mi = blockIdx.x;
  if (mi % a.Q >= a.count[mi / a.Q]) return;                // whole workgroup, before any barrier
  const int H = a.H, W = a.W;
  const long HW = (long)H * W;
  FRLE_STAMP(0);
  const long mask_index = (long)(mi / a.Q) * a.Q + a.sel[mi];
  const FrleLayout L = frle_layout(H, W, a.max_runs);
  unsigned long long* Bw = (unsigned long long*)dyn;
  unsigned* B32 = (unsigned*)dyn;                            // bit (f & 31) of B32[f >> 5] = pixel f (row-major) is set
  unsigned short* part = (unsigned short*)(dyn + L.o_part);
  int* coloff = (int*)(dyn + L.o_coloff);
  int* list = (int*)(dyn + L.o_list);
  if (a.bits) {
    const unsigned long long* src = a.bits + mask_index * L.n64;
    for (long i = tid; i < L.n64; i += nthr) Bw[i] = src[i];
    for (long i = L.n64 + tid; i < L.n64 + 2; i += nthr) Bw[i] = 0;
  } else {
    const unsigned char* m = a.masks + mask_index * HW;
    unsigned short* B16 = (unsigned short*)dyn;
    const long n16 = (HW + 15) >> 4;
    if (HW % 16 == 0 && ((uintptr_t)m & 15) == 0) {
#pragma unroll 8
      for (long i = tid; i < n16; i += nthr) {
        const uint4 v = ((const uint4*)m)[i];
        B16[i] = (unsigned short)(rle_bits4(v.x) | (rle_bits4(v.y) << 4) | (rle_bits4(v.z) << 8) | (rle_bits4(v.w) << 12));
      }
    } else {
      for (long i = tid; i < n16; i += nthr) {
        unsigned b = 0;
        for (int k = 0; k < 16; ++k)
          if (16 * i + k < HW && m[16 * i + k]) b |= 1u << k;
        B16[i] = (unsigned short)b;
      }
    }
    for (long i = n16 + tid; i < (L.n64 + 2) * 4; i += nthr) B16[i] = 0;
  }
  for (int i = tid; i < L.S * L.Wp / 2; i += nthr) ((unsigned*)part)[i] = 0;
  __syncthreads();
  FRLE_STAMP(1);
  // 32 columns of row y, from column 32 p on (bits past the row's end are 0)
  auto rowword = [&](int y, int p) -> unsigned {
    const long f0 = (long)y * W + 32L * p;
    const int sh = (int)(f0 & 31);
    const unsigned lo = B32[f0 >> 5], hi = B32[(f0 >> 5) + 1];
    unsigned w = sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
    const int nb = W - 32 * p;
    if (nb < 32) w &= (1u << nb) - 1u;
    return w;
  };
  // the 16 words of T of item (p, chunk sc) (rows past H: 0); row 0 is compared with the LAST row one column to the left (column 0 of the
  // mask: pixel 0 itself, no transition)
  auto item_words = [&](int p, int sc, unsigned (&t)[FRLE_ROWS], int& area, int& miny, int& maxy, unsigned& ored) {
    const int y0 = sc * FRLE_ROWS;
    unsigned prev;
    if (y0 == 0) {
      const unsigned last = rowword(H - 1, p);
      prev = (last << 1) | (p ? rowword(H - 1, p - 1) >> 31 : (rowword(0, 0) & 1u));
    } else prev = rowword(y0 - 1, p);
    const int nb = min(32, W - 32 * p);
    const unsigned cmask = nb < 32 ? (1u << nb) - 1u : ~0u;
#pragma unroll
    for (int i = 0; i < FRLE_ROWS; ++i) {
      const int y = y0 + i;
      unsigned cur = 0;
      t[i] = 0;
      if (y < H) {
        cur = rowword(y, p);
        t[i] = (cur ^ prev) & cmask;
        prev = cur;
        if (cur) { area += __popc(cur); miny = min(miny, y); maxy = y; ored |= cur; }
      }
    }
  };
  // ---- pass 1: per-column transition counts of every (chunk, panel); area, box
  const int nitems = L.P * L.S;
  int area = 0, miny = H, maxy = -1, minx = W, maxx = -1;
  unsigned t_first[FRLE_ROWS];                               // the thread's first item's words, kept for pass 2 (most shapes: its only item)
  for (int it = tid; it < nitems; it += nthr) {
    const int p = it % L.P, sc = it / L.P;
    unsigned t[FRLE_ROWS], ored = 0;
    item_words(p, sc, t, area, miny, maxy, ored);
    if (it == tid) {
#pragma unroll
      for (int i = 0; i < FRLE_ROWS; ++i) t_first[i] = t[i];
    }
    if (ored) { minx = min(minx, 32 * p + __ffs((int)ored) - 1); maxx = max(maxx, 32 * p + 31 - __clz((int)ored)); }
    unsigned c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0;         // bit j of (c4 c3 c2 c1 c0) = transitions of column j so far (<= 16)
#pragma unroll
    for (int i = 0; i < FRLE_ROWS; ++i) {
      unsigned carry = t[i], x;
      x = c0 & carry; c0 ^= carry; carry = x;
      x = c1 & carry; c1 ^= carry; carry = x;
      x = c2 & carry; c2 ^= carry; carry = x;
      x = c3 & carry; c3 ^= carry; carry = x;
      c4 ^= carry;
    }
    unsigned any = c0 | c1 | c2 | c3 | c4;
    unsigned short* pc = part + (long)sc * L.Wp + 32 * p;
    while (any) {
      const int j = __ffs((int)any) - 1;
      any &= any - 1;
      pc[j] = (unsigned short)(((c0 >> j) & 1u) | (((c1 >> j) & 1u) << 1) | (((c2 >> j) & 1u) << 2) | (((c3 >> j) & 1u) << 3) | (((c4 >> j) & 1u) << 4));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    area += __shfl_xor(area, o, 64);
    miny = min(miny, __shfl_xor(miny, o, 64)); maxy = max(maxy, __shfl_xor(maxy, o, 64));
    minx = min(minx, __shfl_xor(minx, o, 64)); maxx = max(maxx, __shfl_xor(maxx, o, 64));
  }
  if (lane == 0) { s_red[0][wave] = area; s_red[1][wave] = miny; s_red[2][wave] = maxy; s_red[3][wave] = minx; s_red[4][wave] = maxx; }
  __syncthreads();
  FRLE_STAMP(2);
  // ---- per column: exclusive prefix over the chunks (in place), then a block scan of the column totals -> coloff
  int ctot = 0;
  if (tid < L.Wp) {
    for (int sc = 0; sc < L.S; ++sc) { const int v = part[(long)sc * L.Wp + tid]; part[(long)sc * L.Wp + tid] = (unsigned short)ctot; ctot += v; }
  }
  int inc = ctot;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  int wbase = 0, nt = 0;
  for (int w = 0; w < nw; ++w) { if (w < wave) wbase += s_w[w]; nt += s_w[w]; }
  if (tid < L.Wp) coloff[tid] = wbase + inc - ctot;
  int* info = a.info + (long)mi * 8;
  if (tid == 0) {
    int A = 0, y0 = H, y1 = -1, x0 = W, x1 = -1;
    for (int w = 0; w < nw; ++w) { A += s_red[0][w]; y0 = min(y0, s_red[1][w]); y1 = max(y1, s_red[2][w]); x0 = min(x0, s_red[3][w]); x1 = max(x1, s_red[4][w]); }
    info[2] = x0; info[3] = y0; info[4] = x1; info[5] = y1; info[6] = A; info[7] = nt;
  }
  if (nt > a.max_runs) {
    if (tid == 0) { info[0] = 0; info[1] = -1; }
    return;
  }
  const int lead = (int)(B32[0] & 1u), nc = nt + 1 + lead;
  __syncthreads();
  FRLE_STAMP(3);
  // ---- pass 2: the same threads write their transitions to their places in the list, column by column
  for (int it = tid; it < nitems; it += nthr) {
    const int p = it % L.P, sc = it / L.P, y0 = sc * FRLE_ROWS;
    unsigned t[FRLE_ROWS], ored = 0;
    if (it == tid) {
#pragma unroll
      for (int i = 0; i < FRLE_ROWS; ++i) t[i] = t_first[i];
    } else {
      int d0 = 0, d1 = 0, d2 = 0;
      item_words(p, sc, t, d0, d1, d2, ored);
    }
    unsigned any = 0;
#pragma unroll
    for (int i = 0; i < FRLE_ROWS; ++i) any |= t[i];
    const unsigned short* pc = part + (long)sc * L.Wp + 32 * p;
    const int* co = coloff + 32 * p;
    // per column with a transition: its 16 row bits gathered into one word, then one store per set bit; the NEXT column's place is
    // requested from LDS before this column's stores (the two loads were a third of the pass when they sat in front of every column)
    int j = any ? __ffs((int)any) - 1 : 0;
    int o = co[j] + pc[j];
    while (any) {
      any &= any - 1;
      const int jn = any ? __ffs((int)any) - 1 : j;
      const int on = co[jn] + pc[jn];
      unsigned cw = 0;
#pragma unroll
      for (int i = 0; i < FRLE_ROWS; ++i) cw |= ((t[i] >> j) & 1u) << i;
      const int v = (32 * p + j) * H + y0;
      while (cw) { list[o++] = v + __ffs((int)cw) - 1; cw &= cw - 1; }
      j = jn; o = on;
    }
  }
  __syncthreads();
  FRLE_STAMP(4);
  // ---- the string (see mask_rle_kernel) into LDS, over the dead bits / counts
  unsigned char* cbuf = dyn;
  auto run = [&](int k) -> long {
    if (lead) { if (k == 0) return 0; --k; }
    const long lo = k == 0 ? 0 : list[k - 1], hi = k == nt ? HW : list[k];
    return hi - lo;
  };
  auto put = [&](unsigned char* q, long x) {
    bool more = true;
    while (more) {
      long ch = x & 0x1f;
      x >>= 5;
      more = (ch & 0x10) ? x != -1 : x != 0;
      if (more) ch |= 0x20;
      *q++ = (unsigned char)(ch + 48);
    }
  };
  int base = 0;
  for (int k0 = 0; k0 < nc; k0 += 2 * nthr) {                 // two consecutive runs per thread and step (half the barriers)
    const int k = k0 + 2 * tid;
    long xa = 0, xb = 0;
    int na = 0, nb = 0;
    if (k < nc) { xa = run(k); if (k > 2) xa -= run(k - 2); na = rle_groups(xa); }
    if (k + 1 < nc) { xb = run(k + 1); if (k + 1 > 2) xb -= run(k - 1); nb = rle_groups(xb); }
    int in2 = na + nb;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(in2, d, 64); if (lane >= d) in2 += t; }
    __syncthreads();                                        // (the previous step's s_w reads are done)
    if (lane == 63) s_w[wave] = in2;
    __syncthreads();
    int wb = 0, total = 0;
    for (int w = 0; w < nw; ++w) { if (w < wave) wb += s_w[w]; total += s_w[w]; }
    unsigned char* q = cbuf + base + wb + in2 - na - nb;
    if (k < nc) put(q, xa);
    if (k + 1 < nc) put(q + na, xb);
    base += total;
  }
  if (tid == 0) {
    const int off = atomicAdd(a.cursor, base);
    const bool fits = (long)off + base <= a.out_cap;
    info[0] = off; info[1] = fits ? base : -1;
    s_off = fits ? off : -1;
  }
  __syncthreads();
  FRLE_STAMP(5);
  if (s_off < 0) return;
  unsigned char* o = a.out + s_off;
  for (int i = tid; i < base; i += nthr) o[i] = cbuf[i];
  FRLE_STAMP(6);
}
static size_t rle_fused_lds(int H, int W, int max_runs) { return frle_layout(H, W, max_runs).total; }
extern "C" int zh_mask_rle_fused_supported(int H, int W, int max_runs) {
  return H > 0 && W > 0 && W <= 1024 && max_runs > 0 && (long)H * W < (1L << 31) && rle_fused_lds(H, W, max_runs) <= 150 * 1024;
}
extern "C" int zh_mask_rle_fused_kept(const unsigned char* masks, const unsigned long long* bits, const int* kept_index, const int* kept_count,
                                      int B, int Q, int H, int W, int max_runs, unsigned char* out, long out_capacity, int* cursor, int* info,
                                      hipStream_t stream) {
  ZH_CHECK_ARG((masks || bits) && kept_index && kept_count && out && cursor && info && B > 0 && Q > 0 && out_capacity > 0 && (long)B * Q < (1L << 31),
               "zh_mask_rle_fused_kept: bad arguments");
  ZH_CHECK_ARG(zh_mask_rle_fused_supported(H, W, max_runs), "zh_mask_rle_fused_kept: H=%d W=%d max_runs=%d not supported (W <= 1024, bits + tables + list <= 150 KB of LDS)",
               H, W, max_runs);
  FusedRleArgs a{masks, bits, kept_index, kept_count, Q, H, W, max_runs, out, out_capacity, cursor, info};
  const size_t lds = rle_fused_lds(H, W, max_runs);
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)mask_rle_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(mask_rle_fused_kernel, dim3(B * Q), dim3(((W + 63) / 64) * 64), lds, stream, a);     // >= one thread per column (W <= 1024)
  ZH_CHECK_LAUNCH("zh_mask_rle_fused_kept");
  return ZH_OK;
}

// ---- greedy per-category mask NMS on the device (networks/zutis.py:211-299, copy at coco20k_eval.py:54-136).
// One workgroup per image; inputs are the exact integer intersection / union counts of zh_mask_iou_counts, so every IoU is
// inter / (union + 1e-7) in float64 exactly as utils/iou.py:30-32 computes it on boolean masks.  Control flow of the
// reference, restated over the Q x Q counts:
//   for every category present, ascending, skipping 0 (background):
//     active = queries of that category, s = their scores
//     while active: best = argmax s (the reference takes the LAST element of an ascending np.argsort; equal maxima: the
//                   largest query index here — numpy's choice among exact ties is unspecified); emit (best, s[best]);
//                   for the others: s *= w(IoU(m, best)), w = hard: 0 if IoU > thr else 1; linear: (1 - IoU) if IoU > thr
//                   else 1; gaussian: exp(-IoU^2 / sigma); drop those with s <= score_threshold
//     emitted entries whose mask is empty are skipped (zutis.py:281-282).
// Scores are carried in float64: the reference's float32 scores are promoted by the first float64 weight (linear / gaussian)
// and stay float32 under hard NMS, where the only products are x1 / x0 — both representations agree exactly.
#define NMS_MAXQ 1024
__global__ __launch_bounds__(256) void mask_nms_kernel(const int* inter, const int* uni, const float* scores, const long long* cats,
                                                       int Q, int nms_type, double thr, double sigma, double score_thr,
                                                       int* out_idx, double* out_score, long long* out_cat, int* out_count,
                                                       double* packed, const int* range_flag, int* zero_word) {
  __shared__ double s_sc[NMS_MAXQ];
  __shared__ long long s_cat[NMS_MAXQ];
  __shared__ unsigned char s_act[NMS_MAXQ];
  __shared__ double r_val[256];
  __shared__ long long r_key[256];
  __shared__ int s_best, s_n;
  __shared__ long long s_cur;
  const int img = blockIdx.x, tid = threadIdx.x;
  if (zero_word && img == 0 && tid == 0) *zero_word = 0;     // (the cursor of the run / string kernel launched behind this one)
  inter += (long)img * Q * Q; uni += (long)img * Q * Q;
  scores += (long)img * Q; cats += (long)img * Q;
  out_idx += (long)img * Q; out_score += (long)img * Q; out_cat += (long)img * Q;
  for (int q = tid; q < Q; q += 256) { s_cat[q] = cats[q]; s_act[q] = 0; }
  if (tid == 0) { s_n = 0; s_cur = 0; }        // categories <= 0 are never processed (0 = background)
  __syncthreads();
  for (;;) {
    // next category: the smallest id greater than the one just processed
    long long mine = 0x7FFFFFFFFFFFFFFFll;
    for (int q = tid; q < Q; q += 256)
      if (s_cat[q] > s_cur && s_cat[q] < mine) mine = s_cat[q];
    r_key[tid] = mine;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
      if (tid < st && r_key[tid + st] < r_key[tid]) r_key[tid] = r_key[tid + st];
      __syncthreads();
    }
    const long long cur = r_key[0];
    __syncthreads();
    if (cur == 0x7FFFFFFFFFFFFFFFll) break;
    if (tid == 0) s_cur = cur;
    for (int q = tid; q < Q; q += 256) {
      s_act[q] = s_cat[q] == cur;
      s_sc[q] = (double)scores[q];
    }
    __syncthreads();
    for (;;) {
      // argmax over the active set; key = (score, index) so equal maxima resolve to the largest index
      double bv = -1.0; int bi = -1;
      for (int q = tid; q < Q; q += 256)
        if (s_act[q] && (s_sc[q] > bv || (s_sc[q] == bv && q > bi))) { bv = s_sc[q]; bi = q; }
      r_val[tid] = bv; r_key[tid] = bi;
      __syncthreads();
      for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) {
          const double ov = r_val[tid + st]; const long long oi = r_key[tid + st];
          if (oi >= 0 && (r_key[tid] < 0 || ov > r_val[tid] || (ov == r_val[tid] && oi > r_key[tid]))) { r_val[tid] = ov; r_key[tid] = oi; }
        }
        __syncthreads();
      }
      const int best = (int)r_key[0];
      const double best_s = r_val[0];
      __syncthreads();
      if (best < 0) break;
      if (tid == 0) {
        s_act[best] = 0;
        if (inter[(long)best * Q + best] > 0) {         // area of the mask = |m & m|: empty masks are not emitted
          const int n = s_n++;
          out_idx[n] = best; out_score[n] = best_s; out_cat[n] = cur;
        }
      }
      __syncthreads();
      for (int q = tid; q < Q; q += 256) {
        if (!s_act[q]) continue;
        const double iou = (double)inter[(long)q * Q + best] / ((double)uni[(long)q * Q + best] + 1e-7);
        double s = s_sc[q];
        if (nms_type == 0) { if (iou > thr) s = s * 0.0; }
        else if (nms_type == 1) { if (iou > thr) s = s * (1.0 - iou); }
        else s = s * exp(-(iou * iou) / sigma);
        s_sc[q] = s;
        if (!(s > score_thr)) s_act[q] = 0;
      }
      __syncthreads();
    }
  }
  if (tid == 0) out_count[img] = s_n;
  // packed (optional): everything the host needs from this image in ONE row of float64 — [index | score | category | every query's
  // category | count, range flag] (small integers are exact in float64) — so that one device -> host copy fetches it
  if (packed) {
    __syncthreads();
    double* row = packed + (long)img * (4 * Q + 2);
    const int n = s_n;
    for (int q = tid; q < Q; q += 256) {
      row[q] = q < n ? (double)out_idx[q] : -1.0;
      row[Q + q] = q < n ? out_score[q] : 0.0;
      row[2 * Q + q] = q < n ? (double)out_cat[q] : 0.0;
      row[3 * Q + q] = (double)s_cat[q];
    }
    if (tid == 0) { row[4 * Q] = (double)n; row[4 * Q + 1] = range_flag ? (double)*range_flag : 0.0; }
  }
}

// The same loop for Q <= NMS_WAVE_MAXQ candidates with ONE wave per image doing the selection (round 4).  The block-wide form above
// spends its time in __syncthreads: ~11 per selection step (an 8-level argmax tree + 3), ~100 steps for the COCO-20K fixture: 52 us for
// 100 candidates.  Here the whole workgroup first turns the image's integer counts into the Q x Q float64 IoU table in LDS (one
// division per pair, done once, in parallel: the table replaces two loads and a float64 division per candidate and step) and notes which
// masks are empty; then wave 0 alone runs the loop: a lane owns candidates lane, lane + 64, the argmax is a wave reduction of
// (score, index) keys (nms_wave_argmax below), no barrier.  Same arithmetic (float64 scores, IoU = inter / (union + 1e-7) in float64), same tie rule (largest
// index among equal maxima), same emission order.
#define NMS_WAVE_MAXQ 128
// argmax of (score, index) keys over a wave: four DPP steps inside each row of 16 lanes (quad xor 1, quad xor 2, half-row mirror, row
// mirror: both lanes of a pair take the same winner, so after them every lane holds its row's), then the four rows' winners by
// v_readlane — ~40 VALU instructions instead of six rounds of three ds_bpermute (each a trip through the LDS crossbar).
__device__ __forceinline__ void nms_take(double& bv, int& bi, double ov, int oi) {
  if (oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi > bi))) { bv = ov; bi = oi; }
}
template <int CTRL>
__device__ __forceinline__ void nms_dpp_step(double& bv, int& bi) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, bv);
  const int lo = (int)(unsigned)u, hi = (int)(unsigned)(u >> 32);
  const int olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  const int ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  const int oi = __builtin_amdgcn_update_dpp(bi, bi, CTRL, 0xf, 0xf, false);
  nms_take(bv, bi, __builtin_bit_cast(double, ((unsigned long long)(unsigned)ohi << 32) | (unsigned)olo), oi);
}
__device__ __forceinline__ void nms_wave_argmax(double& bv, int& bi) {
  nms_dpp_step<0xB1>(bv, bi);                                 // quad_perm [1,0,3,2]
  nms_dpp_step<0x4E>(bv, bi);                                 // quad_perm [2,3,0,1]
  nms_dpp_step<0x141>(bv, bi);                                // row_half_mirror
  nms_dpp_step<0x140>(bv, bi);                                // row_mirror
  const unsigned long long u = __builtin_bit_cast(unsigned long long, bv);
  const int lo = (int)(unsigned)u, hi = (int)(unsigned)(u >> 32);
  double rv[4]; int ri[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const unsigned l = (unsigned)__builtin_amdgcn_readlane(lo, 16 * r), h = (unsigned)__builtin_amdgcn_readlane(hi, 16 * r);
    rv[r] = __builtin_bit_cast(double, ((unsigned long long)h << 32) | l);
    ri[r] = __builtin_amdgcn_readlane(bi, 16 * r);
  }
  bv = rv[0]; bi = ri[0];
#pragma unroll
  for (int r = 1; r < 4; ++r) nms_take(bv, bi, rv[r], ri[r]);
}
__global__ __launch_bounds__(256) void mask_nms_wave_kernel(const int* inter, const int* uni, const float* scores, const long long* cats,
                                                            int Q, int nms_type, double thr, double sigma, double score_thr,
                                                            int* out_idx, double* out_score, long long* out_cat, int* out_count,
                                                            double* packed, const int* range_flag, int* zero_word) {
  extern __shared__ double s_iou[];                          // [Q][Q] IoU, then [Q] scores of the kept, [Q] (as int) indices / categories
  __shared__ unsigned char s_empty[NMS_WAVE_MAXQ];
  const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  if (zero_word && img == 0 && tid == 0) *zero_word = 0;     // (the cursor of the run / string kernel launched behind this one)
  inter += (long)img * Q * Q; uni += (long)img * Q * Q;
  scores += (long)img * Q; cats += (long)img * Q;
  out_idx += (long)img * Q; out_score += (long)img * Q; out_cat += (long)img * Q;
#pragma unroll 8                                             // eight pairs of loads in flight (40 dependent round trips for Q = 100 otherwise: ~25 of this kernel's 37 us)
  for (int i = tid; i < Q * Q; i += 256) s_iou[i] = (double)inter[i] / ((double)uni[i] + 1e-7);
  for (int q = tid; q < Q; q += 256) s_empty[q] = inter[q * Q + q] <= 0;       // area of the mask = |m & m|: empty masks are not emitted
  __syncthreads();
  if (tid >= 64) return;                                      // wave 0 selects
  constexpr int PL = NMS_WAVE_MAXQ / 64;                     // candidates per lane
  long long cat[PL];
  double sc[PL];
  bool act[PL];
  double sc0[PL];       // the candidates' own scores, read once (they were a dependent global load in front of every category's loop: ~1.5 us x 9)
#pragma unroll
  for (int e = 0; e < PL; ++e) { const int q = lane + 64 * e; cat[e] = q < Q ? cats[q] : 0; sc0[e] = q < Q ? (double)scores[q] : 0.0; }
  int n_out = 0;
  long long cur = 0;                                          // categories <= 0 are never processed (0 = background)
  for (;;) {
    long long mine = 0x7FFFFFFFFFFFFFFFll;
#pragma unroll
    for (int e = 0; e < PL; ++e)
      if (cat[e] > cur && cat[e] < mine) mine = cat[e];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const long long t = __shfl_xor(mine, o, 64); mine = t < mine ? t : mine; }
    if (mine == 0x7FFFFFFFFFFFFFFFll) break;
    cur = mine;
#pragma unroll
    for (int e = 0; e < PL; ++e) { const int q = lane + 64 * e; act[e] = q < Q && cat[e] == cur; sc[e] = sc0[e]; }
    for (;;) {
      double bv = -1.0; int bi = -1;                          // key = (score, index): equal maxima resolve to the largest index
#pragma unroll
      for (int e = 0; e < PL; ++e) {
        const int q = lane + 64 * e;
        if (act[e] && (sc[e] > bv || (sc[e] == bv && q > bi))) { bv = sc[e]; bi = q; }
      }
      nms_wave_argmax(bv, bi);
      if (bi < 0) break;
      const int best = bi;
      if (!s_empty[best]) {
        if (lane == 0) { out_idx[n_out] = best; out_score[n_out] = bv; out_cat[n_out] = cur; }
        ++n_out;
      }
#pragma unroll
      for (int e = 0; e < PL; ++e) {
        const int q = lane + 64 * e;
        if (q == best) act[e] = false;
        if (!act[e]) continue;
        const double iou = s_iou[q * Q + best];
        double v = sc[e];
        if (nms_type == 0) { if (iou > thr) v = v * 0.0; }
        else if (nms_type == 1) { if (iou > thr) v = v * (1.0 - iou); }
        else v = v * exp(-(iou * iou) / sigma);
        sc[e] = v;
        if (!(v > score_thr)) act[e] = false;
      }
    }
  }
  if (lane == 0) out_count[img] = n_out;
  if (packed) {
    __builtin_amdgcn_s_waitcnt(0);                           // lane 0's global stores above, re-read below by the other lanes of this wave
    __threadfence_block();
    double* row = packed + (long)img * (4 * Q + 2);
    for (int q = lane; q < Q; q += 64) {
      row[q] = q < n_out ? (double)out_idx[q] : -1.0;
      row[Q + q] = q < n_out ? out_score[q] : 0.0;
      row[2 * Q + q] = q < n_out ? (double)out_cat[q] : 0.0;
      row[3 * Q + q] = (double)cats[q];
    }
    if (lane == 0) { row[4 * Q] = (double)n_out; row[4 * Q + 1] = range_flag ? (double)*range_flag : 0.0; }
  }
}

extern "C" int zh_mask_nms(const int* inter, const int* uni, const float* scores, const long long* category_ids, int B, int Q,
                           int nms_type, double nms_threshold, double sigma, double score_threshold,
                           int* out_index, double* out_score, long long* out_category, int* out_count, double* packed, const int* range_flag,
                           int* zero_word, hipStream_t stream) {
  ZH_CHECK_ARG(inter && uni && scores && category_ids && out_index && out_score && out_category && out_count,
               "zh_mask_nms: null pointer");
  ZH_CHECK_ARG(B > 0 && Q > 0 && Q <= NMS_MAXQ, "zh_mask_nms: need 0 < Q <= %d", NMS_MAXQ);
  ZH_CHECK_ARG(nms_type >= 0 && nms_type <= 2, "zh_mask_nms: nms_type %d not in {0 hard, 1 linear, 2 gaussian}", nms_type);
  if (Q <= NMS_WAVE_MAXQ) {
    const size_t lds = (size_t)Q * Q * sizeof(double);
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)mask_nms_wave_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(mask_nms_wave_kernel, dim3(B), dim3(256), lds, stream, inter, uni, scores, category_ids, Q, nms_type, nms_threshold, sigma,
                       score_threshold, out_index, out_score, out_category, out_count, packed, range_flag, zero_word);
  } else
  hipLaunchKernelGGL(mask_nms_kernel, dim3(B), dim3(256), 0, stream, inter, uni, scores, category_ids, Q, nms_type, nms_threshold, sigma,
                     score_threshold, out_index, out_score, out_category, out_count, packed, range_flag, zero_word);
  ZH_CHECK_LAUNCH("zh_mask_nms");
  return ZH_OK;
}
