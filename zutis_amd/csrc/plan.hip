// Native launch-plan executor: replays a recorded sequence of C-ABI calls (include/zutis_hip.h) in one C loop, optionally
// two plans round-robin on two streams.  The per-op dispatch is generated from the header (plan_gen.inc).
#include "common.h"
#include "../../include/zutis_hip.h"
#include <string.h>

struct ZhCmd { int op; int pad; unsigned long long a[32]; };

static inline float zh_w2f(unsigned long long w) { unsigned u = (unsigned)w; float f; memcpy(&f, &u, 4); return f; }
static inline double zh_w2d(unsigned long long w) { double d; memcpy(&d, &w, 8); return d; }

static int zh_dispatch(const ZhCmd& c, zh_stream_t stream) {
  switch (c.op) {
#define ZH_PLAN_CASES
#include "plan_gen.inc"
#undef ZH_PLAN_CASES
    default: zh_set_error("zh_plan_run: unknown op %d", c.op); return ZH_ERR_ARG;
  }
}

static const char* const zh_plan_names[] = {
#define ZH_PLAN_NAMES
#include "plan_gen.inc"
#undef ZH_PLAN_NAMES
};

extern "C" const char* zh_plan_op_name(int op) {
  const int n = (int)(sizeof(zh_plan_names) / sizeof(zh_plan_names[0]));
  return (op >= 0 && op < n) ? zh_plan_names[op] : nullptr;
}

extern "C" int zh_plan_run(const void* cmds, int n, zh_stream_t stream) {
  const ZhCmd* c = (const ZhCmd*)cmds;
  for (int i = 0; i < n; ++i) {
    const int rc = zh_dispatch(c[i], stream);
    if (rc != ZH_OK) return rc;
  }
  return ZH_OK;
}

extern "C" int zh_plan_run2(const void* cmds_a, int na, zh_stream_t stream_a, const void* cmds_b, int nb, zh_stream_t stream_b) {
  const ZhCmd* a = (const ZhCmd*)cmds_a;
  const ZhCmd* b = (const ZhCmd*)cmds_b;
  for (int i = 0; i < na || i < nb; ++i) {
    if (i < na) { const int rc = zh_dispatch(a[i], stream_a); if (rc != ZH_OK) return rc; }
    if (i < nb) { const int rc = zh_dispatch(b[i], stream_b); if (rc != ZH_OK) return rc; }
  }
  return ZH_OK;
}

extern "C" int zh_plan_run_multi(const void* const* cmds, const int* n, const zh_stream_t* streams, int count) {
  ZH_CHECK_ARG(cmds && n && streams && count > 0, "zh_plan_run_multi: bad arguments");
  int longest = 0;
  for (int j = 0; j < count; ++j) longest = n[j] > longest ? n[j] : longest;
  for (int i = 0; i < longest; ++i)
    for (int j = 0; j < count; ++j)
      if (i < n[j]) { const int rc = zh_dispatch(((const ZhCmd*)cmds[j])[i], streams[j]); if (rc != ZH_OK) return rc; }
  return ZH_OK;
}
