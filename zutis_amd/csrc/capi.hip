// C-ABI plumbing for libzutis_hip: version + thread-local error text.
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void zh_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* zh_last_error(void) { return g_err; }
extern "C" int zh_version(void) { return 100; }
extern "C" const char* zh_arch(void) { return "gfx950"; }
