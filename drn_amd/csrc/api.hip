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

static int g_tune[16] = {4096, 1, 256, 0, 0, 0, 0, 0, 1, 1, 0, 16, 160, 512, 2048, 1};   // DRN_TUNE_TN3_MINROWS, DRN_TUNE_TN_FUSED, DRN_TUNE_NT_DEEP, DRN_TUNE_EXP0..4, DRN_TUNE_NT_W4, DRN_TUNE_NT_W4C
int drn_tuning(int key) { return g_tune[key]; }
extern "C" int drn_tune(const char* key, int value) {
  drn_clear_status();
  if (key && !strcmp(key, "tn3_minrows")) { g_tune[DRN_TUNE_TN3_MINROWS] = value; return DRN_OK; }
  if (key && !strcmp(key, "tn_fused")) { g_tune[DRN_TUNE_TN_FUSED] = value; return DRN_OK; }
  if (key && !strcmp(key, "nt_w4c")) { g_tune[DRN_TUNE_NT_W4C] = value; return DRN_OK; }
  if (key && !strcmp(key, "nt_w4")) { g_tune[DRN_TUNE_NT_W4] = value; return DRN_OK; }
  if (key && !strcmp(key, "nt_deep")) { g_tune[DRN_TUNE_NT_DEEP] = value; return DRN_OK; }
  if (key && !strcmp(key, "nt_w4h")) { g_tune[DRN_TUNE_NT_W4H] = value; return DRN_OK; }
  if (key && !strcmp(key, "w4h_tapil")) { g_tune[DRN_TUNE_W4H_TAPIL] = value; return DRN_OK; }
  if (key && !strcmp(key, "w4h_halo")) { g_tune[DRN_TUNE_W4H_HALO] = value; return DRN_OK; }
  if (key && !strcmp(key, "bn1_maxwg")) { g_tune[DRN_TUNE_BN1_MAXWG] = value; return DRN_OK; }
  if (key && !strcmp(key, "nt_deep2")) { g_tune[DRN_TUNE_NT_DEEP2] = value; return DRN_OK; }
  if (key && !strcmp(key, "nt_deep_ks")) { g_tune[DRN_TUNE_NT_DEEP_KS] = value; return DRN_OK; }
  if (key && !strncmp(key, "exp", 3) && key[3] >= '0' && key[3] <= '4' && !key[4]) { g_tune[DRN_TUNE_EXP0 + (key[3] - '0')] = value; return DRN_OK; }
  drn_set_error("drn_tune: unknown key %s", key ? key : "(null)");
  return DRN_ERR_ARG;
}
