// Per-row top-k selection for the retrieval step of the index-dataset pipeline (datasets/index_dataset.py:163-167):
// the reference argsorts every category's N similarities (N ~ 2.7 M) and keeps the first 500.  Here: one workgroup per
// row does an exact radix SELECT of the k-th largest (score, index) pair — 8 histogram passes over the row, no sort of N
// elements — then compacts the k survivors and bitonic-sorts only those in LDS.  Ordering: score descending, ties by
// ascending index (torch.argsort(descending=True) leaves tie order unspecified).  HBM-bound: 9 reads of the row.
// Reported index of column i: idx_map ? idx_map[row*ld + i] : i + idx_add — chunk offsets and the merge of per-chunk /
// per-rank candidates (whose columns carry global image indices) need no separate gather.  Ties are always broken by
// ascending COLUMN, so candidate lists must be laid out in ascending-index order of equal scores (chunk-major / rank-major
// concatenations of sorted lists are).  Outputs are rows of stride out_ld (a column block of a wider candidate table).
#include "common.h"

typedef unsigned long long u64;

__device__ __forceinline__ u64 topk_key(float v, unsigned idx) {
  unsigned u = __float_as_uint(v);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);           // unsigned order == float order (NaN sorts above +inf)
  return ((u64)u << 32) | (u64)(0xFFFFFFFFu - idx);         // equal scores: smaller index = larger key
}

#define TOPK_MAXK 1024

__global__ __launch_bounds__(256) void topk_rows_kernel(const float* scores, long ld, long N, int k, const long long* idx_map,
                                                        long long idx_add, long long* idx_out, float* val_out, long out_ld) {
  __shared__ unsigned hist[256];
  __shared__ u64 sel[TOPK_MAXK];
  __shared__ u64 s_prefix;
  __shared__ int s_need, s_count;
  const float* row = scores + (long)blockIdx.x * ld;
  const int tid = threadIdx.x;
  u64 prefix = 0;                       // decided high bytes of the k-th largest key
  int need = k;                         // rank still to locate inside the current prefix bucket
  for (int pass = 0; pass < 8; ++pass) {
    const int shift = 56 - 8 * pass;
    for (int i = tid; i < 256; i += 256) hist[i] = 0;
    __syncthreads();
    const u64 himask = pass == 0 ? 0ull : (~0ull << (shift + 8));
    for (long i = tid; i < N; i += 256) {
      const u64 key = topk_key(row[i], (unsigned)i);
      if ((key & himask) == prefix) atomicAdd(&hist[(unsigned)(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      int acc = 0, b = 255;
      for (; b >= 0; --b) {             // walk buckets from the largest byte down
        if (acc + (int)hist[b] >= need) break;
        acc += (int)hist[b];
      }
      s_prefix = prefix | ((u64)(unsigned)b << shift);
      s_need = need - acc;
    }
    __syncthreads();
    prefix = s_prefix;
    need = s_need;
    __syncthreads();
  }
  // prefix is now exactly the k-th largest key (keys are unique): keep everything >= it
  if (tid == 0) s_count = 0;
  for (int i = tid; i < TOPK_MAXK; i += 256) sel[i] = 0ull;
  __syncthreads();
  for (long i = tid; i < N; i += 256) {
    const u64 key = topk_key(row[i], (unsigned)i);
    if (key >= prefix) { const int slot = atomicAdd(&s_count, 1); if (slot < TOPK_MAXK) sel[slot] = key; }
  }
  __syncthreads();
  // bitonic sort, descending, over the next power of two >= k (padding keys are 0 = smallest)
  int P = 1;
  while (P < k) P <<= 1;
  for (int sz = 2; sz <= P; sz <<= 1)
    for (int st = sz >> 1; st > 0; st >>= 1) {
      for (int i = tid; i < P; i += 256) {
        const int j = i ^ st;
        if (j > i) {
          const bool desc = (i & sz) == 0;
          const u64 a = sel[i], b = sel[j];
          if (desc ? a < b : a > b) { sel[i] = b; sel[j] = a; }
        }
      }
      __syncthreads();
    }
  for (int i = tid; i < k; i += 256) {
    const u64 key = sel[i];
    const unsigned idx = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
    idx_out[(long)blockIdx.x * out_ld + i] = idx_map ? idx_map[(long)blockIdx.x * ld + idx] : (long long)idx + idx_add;
    if (val_out) val_out[(long)blockIdx.x * out_ld + i] = row[idx];
  }
}

extern "C" int zh_topk_rows(const float* scores, long ld, int rows, long N, int k, const long long* idx_map, long long idx_add,
                            long long* idx_out, float* val_out, long out_ld, hipStream_t stream) {
  ZH_CHECK_ARG(scores && idx_out && rows > 0 && N > 0 && k > 0 && out_ld >= k, "zh_topk_rows: bad arguments");
  ZH_CHECK_ARG(k <= TOPK_MAXK && k <= N, "zh_topk_rows: k=%d must be <= min(%d, N)", k, TOPK_MAXK);
  ZH_CHECK_ARG(N < 4294967295L && ld >= N, "zh_topk_rows: N must fit 32 bits and ld >= N");
  hipLaunchKernelGGL(topk_rows_kernel, dim3(rows), dim3(256), 0, stream, scores, ld, N, k, idx_map, idx_add, idx_out, val_out,
                     out_ld);
  ZH_CHECK_LAUNCH("zh_topk_rows");
  return ZH_OK;
}
