// Error reporting + version of libdvis_hip.so.
#include "dvis_common.h"

static thread_local char g_err[512] = "";

void dvis_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

DVIS_EXPORT const char *dvis_last_error(void) { return g_err; }
DVIS_EXPORT int dvis_version(void) { return 100; }
