// Evaluation metrics on device: confusion-matrix histogram (utils/running_score.py:11-16).
#include "common.h"

// hist[n*gt + pred] += 1 for 0 <= gt < n.  LDS-privatised when n*n fits (81 classes -> 26 KiB),
// otherwise straight global integer atomics (920 classes).  Integer adds => order-independent, exact.
#define HIST_LDS_MAX 12288

__global__ __launch_bounds__(256) void confusion_hist_kernel(const long long* gt, const long long* pred, unsigned long long* hist,
                                                             long total, int n, int use_lds) {
  __shared__ unsigned int sh[HIST_LDS_MAX];
  const int nn = n * n;
  if (use_lds) {
    for (int i = threadIdx.x; i < nn; i += 256) sh[i] = 0;
    __syncthreads();
  }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long long g = gt[i];
    if (g >= 0 && g < n) {
      const long long pr = pred[i];
      if (pr >= 0 && pr < n) {
        const int k = (int)(n * g + pr);
        if (use_lds) atomicAdd(&sh[k], 1u);
        else atomicAdd(&hist[k], 1ull);
      }
    }
  }
  if (use_lds) {
    __syncthreads();
    for (int i = threadIdx.x; i < nn; i += 256)
      if (sh[i]) atomicAdd(&hist[i], (unsigned long long)sh[i]);
  }
}

extern "C" int zh_confusion_hist(const long long* label_true, const long long* label_pred, long long* hist_accum, long total,
                                 int n_class, hipStream_t stream) {
  ZH_CHECK_ARG(label_true && label_pred && hist_accum && total > 0 && n_class > 0 && n_class <= 46340, "zh_confusion_hist: bad arguments");
  const int use_lds = n_class * n_class <= HIST_LDS_MAX;
  long blocks = zh_cdiv(total, 256 * 16);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(confusion_hist_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, label_true, label_pred,
                     (unsigned long long*)hist_accum, total, n_class, use_lds);
  ZH_CHECK_LAUNCH("zh_confusion_hist");
  return ZH_OK;
}
