// Library-level entry points of libbevf_hip.so: version and per-thread error text.
#include "common.h"

static thread_local char g_err[512] = "";

void bevf_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int bevf_version(void) { return 220; }  // 0.2.2: round 2 -- decode / conv descriptors grew; Winograd forward + weight gradient, fused stem + pool, fused PointNet front entries
extern "C" const char* bevf_last_error(void) { return g_err; }
