// ABI bookkeeping: version + thread-local error string (include/drn_hip.h).
#include <stdarg.h>
#include "common.h"
#include "../../include/drn_hip.h"

static thread_local char g_err[512] = "";

void drn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int drn_abi_version(void) {
  drn_clear_status(); return DRN_ABI_VERSION; }
extern "C" const char* drn_last_error(void) { return g_err; }
