// Host-side plumbing of libstpde_hip: version, error text, launch checking.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/stpde_hip.h"

static thread_local char g_err[512] = "";

void stpde_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int stpde_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    stpde_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return STPDE_E_LAUNCH;
  }
  return STPDE_OK;
}

extern "C" int stpde_version(void) { return 100; }

extern "C" int stpde_last_error(char* buf, unsigned long n) {
  if (!buf || n == 0) return STPDE_E_BADARG;
  strncpy(buf, g_err, n - 1);
  buf[n - 1] = 0;
  return STPDE_OK;
}
