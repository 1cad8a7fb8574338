// Host-side plumbing of libstpde_hip: version, error text, launch checking.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <set>
#include <string>

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

// ABI version: bumped whenever a descriptor or a signature of include/stpde_hip.h changes (_lib.ABI_VERSION must match)
extern "C" int stpde_version(void) { return 314; }

// ---- launch-geometry overrides for tests (stpde_tune, include/stpde_hip.h) ---------------------------------------
// Not performance switches: the persistent-grid kernels of the U-Net pick their own grid and their own minimum volume; tests
// force other values to reach the multi-block-per-workgroup and ragged-tail paths on volumes a test can afford.
#include <atomic>
static const char* const g_tune_names[STPDE_TUNE_COUNT] = {"conv3_lds_off", "conv3_lds_minblk", "conv3_lds_gx", "conv_wgrad_lds_gx",
                                                           "conv1_wgrad_lds_gx"};
static std::atomic<int> g_tune[STPDE_TUNE_COUNT];
int stpde_tune_get(int key) { return (key >= 0 && key < STPDE_TUNE_COUNT) ? g_tune[key].load(std::memory_order_relaxed) : 0; }
extern "C" int stpde_tune(const char* name, int value) {
  if (!name) return -1;
  for (int k = 0; k < STPDE_TUNE_COUNT; ++k)
    if (!strcmp(name, g_tune_names[k])) return g_tune[k].exchange(value);
  stpde_set_error("stpde_tune: unknown key '%s'", name);
  return -1;
}

extern "C" int stpde_last_error(char* buf, unsigned long n) {
  if (!buf || n == 0) return STPDE_E_BADARG;
  strncpy(buf, g_err, n - 1);
  buf[n - 1] = 0;
  return STPDE_OK;
}

// ---- dispatch trace (tests assert which template instantiations a call reached) -------------------------------
int stpde_trace_on = 0;
static std::mutex g_trace_mu;
static std::set<std::string> g_trace;

void stpde_trace_note(const char* launcher, const char* kernel) {
  std::lock_guard<std::mutex> lk(g_trace_mu);
  g_trace.insert(std::string(kernel) + " @ " + launcher);
}

extern "C" int stpde_trace_enable(int on) {
  std::lock_guard<std::mutex> lk(g_trace_mu);
  g_trace.clear();
  stpde_trace_on = on ? 1 : 0;
  return STPDE_OK;
}

extern "C" long stpde_trace_read(char* buf, unsigned long n) {
  std::lock_guard<std::mutex> lk(g_trace_mu);
  std::string all;
  for (const auto& s : g_trace) all += s + "\n";
  if (buf && n > 0) {
    strncpy(buf, all.c_str(), n - 1);
    buf[n - 1] = 0;
  }
  return (long)all.size() + 1;   // bytes needed
}
