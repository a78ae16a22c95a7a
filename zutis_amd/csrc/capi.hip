// C-ABI plumbing for libzutis_hip: version + thread-local error text.
#include "common.h"
#include "../../include/zutis_hip.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void zh_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* zh_last_error(void) { return g_err; }
extern "C" int zh_version(void) { return ZH_ABI_VERSION; }   // include/zutis_hip.h
extern "C" const char* zh_arch(void) { return "gfx950"; }

// ---- developer overrides of the GEMM tile choice (gemm_kernel.h): one process-wide instance, initialised from the environment
//      once, changed only through zh_dev_set_gemm_overrides (tests force every tile variant with it)
#include <stdlib.h>
struct GemmDevOverrides { int group_m; int tile; int tile_small; };
static GemmDevOverrides& gemm_dev_state() {
  static GemmDevOverrides o = [] {
    GemmDevOverrides v{4, 0, 0};
    if (const char* g = getenv("ZH_GEMM_GROUP_M")) { const int x = atoi(g); if (x >= 1) v.group_m = x; }
    if (const char* t = getenv("ZH_GEMM_TILE")) v.tile = atoi(t);
    if (const char* t = getenv("ZH_GEMM_TILE_SMALL")) v.tile_small = atoi(t);
    return v;
  }();
  return o;
}
const GemmDevOverrides& gemm_dev_overrides() { return gemm_dev_state(); }
extern "C" int zh_dev_set_gemm_overrides(int group_m, int tile, int tile_small) {
  ZH_CHECK_ARG(group_m >= 0 && tile >= 0 && tile_small >= 0, "zh_dev_set_gemm_overrides: negative argument");
  GemmDevOverrides& o = gemm_dev_state();
  o.group_m = group_m >= 1 ? group_m : 4;
  o.tile = tile;
  o.tile_small = tile_small;
  return ZH_OK;
}

// persistent big-tile GEMMs (gemm_kernel.h PERS): workgroups per launch = the CUs of the CURRENT device rounded down to a multiple
// of 8 (a persistent workgroup takes a CU's whole LDS; the tile walk deals ids modulo 8 to the XCDs) — 256 on an MI355X in SPX mode,
// fewer in a partitioned mode.  Resolved at the first launch (no HIP call at load time: build() loads the library without a GPU) and
// cached per process; ZH_GEMM_PERSIST / zh_dev_set_gemm_persist override it for tests and A/B runs (0 = off, else a multiple of 8).
// The one mutable word that steers product launches: written only by the developer setter, read once per launch.
static int gemm_persist_valid(int n) { return n >= 0 && n % 8 == 0; }
static int g_gemm_persist = [] {
  const char* e = getenv("ZH_GEMM_PERSIST");
  if (!e) return -1;                                    // auto
  char* end = nullptr;
  const long v = strtol(e, &end, 10);
  if (end == e || *end != 0 || v < 0 || v > (1 << 20) || !gemm_persist_valid((int)v)) {
    fprintf(stderr, "zutis_hip: ZH_GEMM_PERSIST=%s ignored (not 0 or a multiple of 8)\n", e);
    return -1;
  }
  return (int)v;
}();
int gemm_persist_cus() {
  if (g_gemm_persist >= 0) return g_gemm_persist;
  static const int auto_cus = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) return 256;
    return cus / 8 * 8;
  }();
  return auto_cus;
}
extern "C" int zh_dev_set_gemm_persist(int workgroups) {
  ZH_CHECK_ARG(workgroups >= -1 && (workgroups == -1 || gemm_persist_valid(workgroups)), "zh_dev_set_gemm_persist: %d is not -1 (auto), 0 (off) or a multiple of 8", workgroups);
  g_gemm_persist = workgroups;
  return ZH_OK;
}

// ---- host-side COCO RLE (pycocotools maskApi.c rleEncode + rleToString restated): the reference encodes every kept
//      instance mask with pycocotools.mask.encode(np.asfortranarray(m)) (networks/zutis.py:290,448).  HOST pointers.
//      mask u8 [H,W] row-major; runs are taken in column-major order starting with the zeros run.  Returns the string
//      length, or -1 if `cap` is too small.
extern "C" long zh_rle_encode_host(const unsigned char* mask, int H, int W, char* out, long cap) {
  long n = 0;
  long cnts_prev2 = 0, cnts_prev1 = 0;     // counts[i-2], counts[i-1]
  long idx = 0;
  auto emit = [&](long c) -> bool {
    long x = c;
    if (idx > 2) x -= cnts_prev2;
    bool more = true;
    while (more) {
      long ch = x & 0x1f;
      x >>= 5;
      more = (ch & 0x10) ? x != -1 : x != 0;
      if (more) ch |= 0x20;
      if (n >= cap) return false;
      out[n++] = (char)(ch + 48);
    }
    cnts_prev2 = cnts_prev1; cnts_prev1 = c; ++idx;
    return true;
  };
  unsigned char cur = 0;
  long run = 0;
  for (int x = 0; x < W; ++x)
    for (int y = 0; y < H; ++y) {
      const unsigned char v = mask[(long)y * W + x] != 0;
      if (v != cur) { if (!emit(run)) return -1; run = 0; cur = v; }
      ++run;
    }
  if (!emit(run)) return -1;
  return n;
}

// HOST: RLE string from run lengths (pycocotools rleToString): counts int64 [n] -> chars; returns length or -1.
extern "C" long zh_rle_counts_to_string_host(const long long* counts, long n, char* out, long cap) {
  long len = 0;
  for (long i = 0; i < n; ++i) {
    long x = (long)counts[i];
    if (i > 2) x -= (long)counts[i - 2];
    bool more = true;
    while (more) {
      long ch = x & 0x1f;
      x >>= 5;
      more = (ch & 0x10) ? x != -1 : x != 0;
      if (more) ch |= 0x20;
      if (len >= cap) return -1;
      out[len++] = (char)(ch + 48);
    }
  }
  return len;
}

// HOST: the RLE strings of n masks straight from zh_mask_runs' output, one call (rle_from_transitions was one numpy diff + one
// ctypes call per mask: 18 us each, 0.3 ms of a 0.7-ms batch-1 instance predict for 17 kept masks).  positions int32 [n, stride]
// (column-major pixel indices where the value changes, the first nruns[2 i] of row i), nruns int32 [n, 2] = {transitions, value of
// pixel 0}; counts = diff([0, positions..., HW]) with a leading empty zero-run when pixel 0 is set (pycocotools starts with zeros).
// Strings are written back to back into `out`; offsets int64 [n + 1].  Masks with more than `stride` transitions get an empty
// string (offsets equal): the caller falls back to the mask encoder.  Returns the total length or -1 when `cap` is too small.
// packed != 0: `positions` is zh_mask_runs_kept's packed list — row i starts where row i - 1 ended, min(transitions, stride) entries each.
extern "C" long zh_rle_from_transitions_host(const int* positions, long stride, int packed, const int* nruns, long n, long HW, char* out,
                                             long cap, long long* offsets) {
  long len = 0, at = 0;
  for (long i = 0; i < n; ++i) {
    offsets[i] = len;
    const long nt = nruns[2 * i];
    const int* p = packed ? positions + at : positions + i * stride;
    at += nt < stride ? nt : stride;
    if (nt > stride) continue;
    const bool lead = nruns[2 * i + 1] != 0;
    const long nc = nt + 1 + (lead ? 1 : 0);              // number of runs
    long c1 = 0, c2 = 0;                                   // counts[k - 1], counts[k - 2]
    for (long k = 0; k < nc; ++k) {
      long cnt;
      if (lead && k == 0) cnt = 0;
      else {
        const long e = k - (lead ? 1 : 0);                 // run e spans [edge(e), edge(e + 1))
        const long lo = e == 0 ? 0 : p[e - 1], hi = e == nt ? HW : p[e];
        cnt = hi - lo;
      }
      long x = cnt;
      if (k > 2) x -= c2;
      bool more = true;
      while (more) {
        long ch = x & 0x1f;
        x >>= 5;
        more = (ch & 0x10) ? x != -1 : x != 0;
        if (more) ch |= 0x20;
        if (len >= cap) return -1;
        out[len++] = (char)(ch + 48);
      }
      c2 = c1; c1 = cnt;
    }
  }
  offsets[n] = len;
  return len;
}

