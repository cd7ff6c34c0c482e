// Error reporting for the C-ABI (thread-local message, no exceptions across the boundary).
#include "../../include/neurosis_hip.h"
#include "nk_common.h"
#include <stdio.h>

static thread_local char g_err[512] = "";

void nk_set_error(const char* file, int line, const char* what) {
  snprintf(g_err, sizeof(g_err), "%s:%d: %s", file, line, what);
}

int nk_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: launch failed: %s", what, hipGetErrorString(e));
    return NK_ERR_LAUNCH;
  }
  return NK_OK;
}

extern "C" const char* nk_last_error(void) { return g_err; }
extern "C" int nk_abi_version(void) { return 1; }
