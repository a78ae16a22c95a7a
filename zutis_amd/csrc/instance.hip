// Instance-prediction kernels (networks/zutis.py:374-420): per-query mask statistics, masked mean of the
// text-space patch tokens, and the per-query class / score.  The reference materialises
// patch_tokens[:,None] * binary_masks[...,None] = B x Q x hw x 512 fp32 (2.9 GB at B=8, zutis.py:404-406);
// here the masked mean is a streaming reduction that reads each image's tokens from L2.
#include "common.h"

// ---- per (image, query): size = #(p > thr), conf = sum(p * (p > thr)) / (size + 1e-7)   (zutis.py:390-397)
// range_flag (optional): bit 0 is set when any proposal lies outside [0, 1] (or is a NaN) — the reference's two asserts on the
// mask proposals (zutis.py:385-386), checked by the host at the predict's one synchronisation instead of with a reduction + copy of its own
__global__ __launch_bounds__(256) void instance_stats_kernel(const float* mp, long stride_b, float thr, long rows, int Q, int M,
                                                             float* sizes, float* conf, unsigned char* binary, int* range_flag) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);       // r = b*Q + q
  if (r >= rows) return;
  const int b = (int)(r / Q), q = (int)(r % Q);
  const float* p = mp + (long)b * stride_b + (long)q * M;
  float cnt = 0.f, s = 0.f;
  bool bad = false;
  for (int m = lane; m < M; m += 64) {
    const float v = p[m];
    bad |= !(v >= 0.f && v <= 1.f);
    const bool on = v > thr;
    cnt += on ? 1.f : 0.f;
    s += on ? v : 0.f;
    binary[r * M + m] = on ? 1 : 0;
  }
  cnt = wave_sum(cnt);
  s = wave_sum(s);
  if (lane == 0) { sizes[r] = cnt; conf[r] = s / (cnt + 1e-7f); }
  if (range_flag && __ballot(bad) != 0ull && lane == 0) atomicOr(range_flag, 1);
}

extern "C" int zh_instance_mask_stats(const float* mask_proposals, long stride_image, float threshold, int B, int Q, int M,
                                      float* sizes, float* confidence, unsigned char* binary, int* range_flag, hipStream_t stream) {
  ZH_CHECK_ARG(mask_proposals && sizes && confidence && binary && B > 0 && Q > 0 && M > 0, "zh_instance_mask_stats: bad arguments");
  const long rows = (long)B * Q;
  hipLaunchKernelGGL(instance_stats_kernel, dim3(zh_cdiv(rows, 4)), dim3(256), 0, stream, mask_proposals, stride_image, threshold, rows, Q, M,
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
#define QT 10
#define MCH 128
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
#pragma unroll 8                                         // eight rows of loads in flight (same summation order per accumulator)
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
  // three classes per pass (cls, cls + 4, cls + 8): their rows are loaded together — one class at a time was 21 dependent passes of
  // load -> reduce -> exp per wave, 48 us for 100 queries x 81 classes at batch 1.  Each class's dot product keeps its summation
  // order and the classes are compared in ascending order, as before: bit-identical categories and scores.
  for (int cls = wave; cls < n; cls += 12) {
    const float* t[3];
    float d[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 3; ++j) t[j] = text + (long)(cls + 4 * j < n ? cls + 4 * j : cls) * E;
    for (int c = lane; c < E; c += 64) {
      const float x = sv[c] * inv;
#pragma unroll
      for (int j = 0; j < 3; ++j) d[j] += t[j][c] * x;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
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

__global__ __launch_bounds__(256) void mask_iou_counts_kernel(const unsigned long long* packed, int n, long W64, int* inter, int* uni) {
  const int i = blockIdx.x, j = blockIdx.y;
  if (j < i) return;
  const unsigned long long* a = packed + (long)i * W64;
  const unsigned long long* b = packed + (long)j * W64;
  int ci = 0, cu = 0;
  for (long w = threadIdx.x; w < W64; w += 256) {
    const unsigned long long x = a[w], y = b[w];
    ci += __popcll(x & y);
    cu += __popcll(x | y);
  }
  __shared__ int ri[4], ru[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { ci += __shfl_xor(ci, o, 64); cu += __shfl_xor(cu, o, 64); }
  if ((threadIdx.x & 63) == 0) { ri[threadIdx.x >> 6] = ci; ru[threadIdx.x >> 6] = cu; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int I = ri[0] + ri[1] + ri[2] + ri[3], U = ru[0] + ru[1] + ru[2] + ru[3];
    inter[(long)i * n + j] = I; inter[(long)j * n + i] = I;
    uni[(long)i * n + j] = U; uni[(long)j * n + i] = U;
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
  hipLaunchKernelGGL(mask_iou_counts_kernel, dim3(n, n), dim3(256), 0, stream, (const unsigned long long*)workspace, n, W64, inter, uni);
  ZH_CHECK_LAUNCH("zh_mask_iou_counts");
  return ZH_OK;
}

// ---- run-length transitions + box + area of selected masks ON THE DEVICE (replaces the B x Q x H x W mask D2H in front of
//      pycocotools.mask.encode / masks_to_boxes, networks/zutis.py:288-294,446-452).  COCO RLE runs are column-major:
//      the mask is cut into 64-column panels staged through LDS (coalesced row reads); the value changes per column segment are
//      counted, prefix-summed, and the column-major pixel positions where the value changes are written.
//      counts = diff([0, positions..., H*W]) with a leading 0-run inserted when pixel 0 is set (host, tiny).
// One block per (mask, 64-column panel) — round 3: one block per mask walked its 480 x 640 pixels with one thread per column, twice,
// 17 blocks on 256 CUs: 194 us, a sixth of the batch-1 instance predict.  A panel's block
//   1. counts the transitions in the columns LEFT of its panel straight from global memory, row-major and coalesced (a vertical
//      transition is a difference between a row and the row above it; plus the H-1 -> 0 wrap between neighbouring columns): the
//      offset of its first run in the mask's position list, no workspace and no second launch;
//   2. stages its panel [H][64] in LDS, four threads per column (a quarter of the rows each) count their segment's transitions,
//      a 256-entry scan in (column, segment) order places them, a second walk writes the column-major positions;
//   3. the block of the LAST panel also knows the total; the block of the FIRST panel (nothing to its left) scans the whole mask once
//      more (row-major) for the box and the area.
#define RUNS_PANEL 64
#define RUNS_SEG 4
__device__ __forceinline__ unsigned zh_nz_bytes(unsigned w) {   // 0x01 in every byte of w that is non-zero
  const unsigned t = ((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w;
  return (t >> 7) & 0x01010101u;
}
// count != NULL (zh_mask_runs_kept): entry mi = b * Q + j is the j-th kept query of image b — sel[mi] is its query index (zh_mask_nms'
// out_index), entries j >= count[b] do nothing: the NMS result never visits the host before the runs are extracted.
__global__ __launch_bounds__(256) void mask_runs_kernel(const unsigned char* masks, const int* sel, const int* count, int Q, int H, int W, int max_runs,
                                                        int* positions, int* nruns, int* box_area) {
  extern __shared__ unsigned char sm[];                    // [H][64] panel
  __shared__ int s_red[256];
  __shared__ int s_minx, s_maxx, s_miny, s_maxy, s_area;
  const int tid = threadIdx.x, mi = blockIdx.x, pnl = blockIdx.y;
  if (count && mi % Q >= count[mi / Q]) return;           // whole workgroup, before any barrier
  const unsigned char* m = masks + (count ? (long)(mi / Q) * Q + sel[mi] : (long)sel[mi]) * H * W;
  int* pos = positions + (long)mi * max_runs;
  const int x0 = pnl * RUNS_PANEL, pw = min(RUNS_PANEL, W - x0);
  const bool vec = (W % 16 == 0) && (((uintptr_t)m & 15) == 0);
  // ---- 1. transitions in columns [0, x0)
  int cnt = 0;
  if (x0 > 0) {
    if (vec) {                                             // x0 % 64 == 0: whole 16-column chunks
      const int cpr = x0 / 16;
#pragma unroll 8
      for (int i = tid; i < (H - 1) * cpr; i += 256) {              // independent iterations: eight pairs of loads in flight
        const int y = 1 + i / cpr, c16 = i % cpr;
        const uint4 a = *(const uint4*)(m + (long)y * W + c16 * 16), b = *(const uint4*)(m + (long)(y - 1) * W + c16 * 16);
        cnt += __popc(zh_nz_bytes(a.x) ^ zh_nz_bytes(b.x)) + __popc(zh_nz_bytes(a.y) ^ zh_nz_bytes(b.y)) +
               __popc(zh_nz_bytes(a.z) ^ zh_nz_bytes(b.z)) + __popc(zh_nz_bytes(a.w) ^ zh_nz_bytes(b.w));
      }
    } else {
      for (long i = tid; i < (long)(H - 1) * x0; i += 256) {
        const int y = 1 + (int)(i / x0), x = (int)(i % x0);
        cnt += (m[(long)y * W + x] != 0) != (m[(long)(y - 1) * W + x] != 0);
      }
    }
    for (int x = 1 + tid; x < x0; x += 256)                // the wrap from the bottom of column x - 1 to the top of column x
      cnt += (m[x] != 0) != (m[(long)(H - 1) * W + x - 1] != 0);
  }
  s_red[tid] = cnt;
  // ---- 2. the panel
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
  for (int st = 128; st > 0; st >>= 1) {                   // block sum of the left-hand count
    if (tid < st) s_red[tid] += s_red[tid + st];
    __syncthreads();
  }
  const int base = s_red[0];
  __syncthreads();
  // thread = (column c, segment sg) in the order the positions are emitted: entry e = c * RUNS_SEG + sg == tid
  const int c = tid / RUNS_SEG, sg = tid % RUNS_SEG;
  const int hs = (H + RUNS_SEG - 1) / RUNS_SEG, y0 = min(H, sg * hs), y1 = min(H, y0 + hs);
  unsigned char prev0 = 0;
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
  s_red[tid] = mine;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {                      // inclusive scan (Hillis - Steele)
    const int t = tid >= d ? s_red[tid - d] : 0;
    __syncthreads();
    s_red[tid] += t;
    __syncthreads();
  }
  const int total = base + s_red[255];
  if (mine) {
    int o = base + s_red[tid] - mine;
    unsigned char prev = prev0;
#pragma unroll 8
    for (int y = y0; y < y1; ++y) {
      const unsigned char v = sm[y * RUNS_PANEL + c];
      if (v != prev) { if (o < max_runs) pos[o] = (x0 + c) * H + y; ++o; }
      prev = v;
    }
  }
  // ---- 3. the total (known to the last panel's block); box and area from one more row-major scan of the whole mask — by the
  //         FIRST panel's block, which had no columns to its left to count (the last one had all of them)
  if (pnl == (int)gridDim.y - 1 && tid == 0) {
    nruns[mi * 2] = total;                                 // number of transitions (may exceed max_runs => host fallback)
    nruns[mi * 2 + 1] = m[0] != 0;                         // value of pixel 0
  }
  if (pnl != 0) return;
  if (tid == 0) { s_minx = W; s_maxx = -1; s_miny = H; s_maxy = -1; s_area = 0; }
  __syncthreads();
  int area = 0, minx = W, maxx = -1, miny = H, maxy = -1;
  if (vec) {
    const int cpr = W / 16;
#pragma unroll 8
    for (int i = tid; i < H * cpr; i += 256) {
      const int y = i / cpr, c16 = i - y * cpr;
      const uint4 v = *(const uint4*)(m + (long)y * W + c16 * 16);
      const unsigned w4[4] = {zh_nz_bytes(v.x), zh_nz_bytes(v.y), zh_nz_bytes(v.z), zh_nz_bytes(v.w)};
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (w4[k]) {
          area += __popc(w4[k]);
          const int xb = c16 * 16 + k * 4;
          minx = min(minx, xb + (__ffs(w4[k]) - 1) / 8);
          maxx = max(maxx, xb + (31 - __clz(w4[k])) / 8);
          miny = min(miny, y); maxy = max(maxy, y);
        }
    }
  } else {
    for (long i = tid; i < (long)H * W; i += 256) {
      if (m[i]) {
        const int y = (int)(i / W), x = (int)(i % W);
        ++area; minx = min(minx, x); maxx = max(maxx, x); miny = min(miny, y); maxy = max(maxy, y);
      }
    }
  }
  if (area) {
    atomicAdd(&s_area, area); atomicMin(&s_minx, minx); atomicMax(&s_maxx, maxx); atomicMin(&s_miny, miny); atomicMax(&s_maxy, maxy);
  }
  __syncthreads();
  if (tid == 0) {
    int* b = box_area + mi * 5;
    b[0] = s_minx; b[1] = s_miny; b[2] = s_maxx; b[3] = s_maxy; b[4] = s_area;
  }
}

extern "C" int zh_mask_runs(const unsigned char* masks, const int* sel, int n_sel, int H, int W, int max_runs,
                            int* positions, int* nruns, int* box_area, hipStream_t stream) {
  ZH_CHECK_ARG(masks && sel && positions && nruns && box_area && n_sel > 0 && H > 0 && W > 0 && max_runs > 0, "zh_mask_runs: bad arguments");
  ZH_CHECK_ARG((long)H * W < (1L << 31), "zh_mask_runs: mask too large");
  const size_t lds = ((size_t)H * RUNS_PANEL + 15) & ~(size_t)15;
  ZH_CHECK_ARG(lds <= 150 * 1024, "zh_mask_runs: H=%d too tall for the LDS panel", H);
  const int npanel = zh_cdiv(W, RUNS_PANEL);
  ZH_CHECK_ARG(npanel <= 65535, "zh_mask_runs: mask too wide");
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)mask_runs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(mask_runs_kernel, dim3(n_sel, npanel), dim3(256), lds, stream, masks, sel, (const int*)nullptr, 1, H, W, max_runs, positions, nruns, box_area);
  ZH_CHECK_LAUNCH("zh_mask_runs");
  return ZH_OK;
}

extern "C" int zh_mask_runs_kept(const unsigned char* masks, const int* kept_index, const int* kept_count, int B, int Q, int H, int W, int max_runs,
                                 int* positions, int* nruns, int* box_area, hipStream_t stream) {
  ZH_CHECK_ARG(masks && kept_index && kept_count && positions && nruns && box_area && B > 0 && Q > 0 && H > 0 && W > 0 && max_runs > 0,
               "zh_mask_runs_kept: bad arguments");
  ZH_CHECK_ARG((long)H * W < (1L << 31) && (long)B * Q < (1L << 31), "zh_mask_runs_kept: mask / batch too large");
  const size_t lds = ((size_t)H * RUNS_PANEL + 15) & ~(size_t)15;
  ZH_CHECK_ARG(lds <= 150 * 1024, "zh_mask_runs_kept: H=%d too tall for the LDS panel", H);
  const int npanel = zh_cdiv(W, RUNS_PANEL);
  ZH_CHECK_ARG(npanel <= 65535, "zh_mask_runs_kept: mask too wide");
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)mask_runs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(mask_runs_kernel, dim3(B * Q, npanel), dim3(256), lds, stream, masks, kept_index, kept_count, Q, H, W, max_runs, positions, nruns, box_area);
  ZH_CHECK_LAUNCH("zh_mask_runs_kept");
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
                                                       double* packed, const int* range_flag) {
  __shared__ double s_sc[NMS_MAXQ];
  __shared__ long long s_cat[NMS_MAXQ];
  __shared__ unsigned char s_act[NMS_MAXQ];
  __shared__ double r_val[256];
  __shared__ long long r_key[256];
  __shared__ int s_best, s_n;
  __shared__ long long s_cur;
  const int img = blockIdx.x, tid = threadIdx.x;
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

extern "C" int zh_mask_nms(const int* inter, const int* uni, const float* scores, const long long* category_ids, int B, int Q,
                           int nms_type, double nms_threshold, double sigma, double score_threshold,
                           int* out_index, double* out_score, long long* out_category, int* out_count, double* packed, const int* range_flag,
                           hipStream_t stream) {
  ZH_CHECK_ARG(inter && uni && scores && category_ids && out_index && out_score && out_category && out_count,
               "zh_mask_nms: null pointer");
  ZH_CHECK_ARG(B > 0 && Q > 0 && Q <= NMS_MAXQ, "zh_mask_nms: need 0 < Q <= %d", NMS_MAXQ);
  ZH_CHECK_ARG(nms_type >= 0 && nms_type <= 2, "zh_mask_nms: nms_type %d not in {0 hard, 1 linear, 2 gaussian}", nms_type);
  hipLaunchKernelGGL(mask_nms_kernel, dim3(B), dim3(256), 0, stream, inter, uni, scores, category_ids, Q, nms_type, nms_threshold, sigma,
                     score_threshold, out_index, out_score, out_category, out_count, packed, range_flag);
  ZH_CHECK_LAUNCH("zh_mask_nms");
  return ZH_OK;
}
