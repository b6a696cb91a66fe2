// Library-level entry points: version, thread-local error string, device check.
#include <stdarg.h>

#include <mutex>

#include "common.h"

static thread_local char g_err[512] = "";

void re2e_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int re2e_version(void) { return RE2E_ABI_VERSION; }

extern "C" const char* re2e_last_error(void) { return g_err; }

extern "C" int re2e_device_ok(void) {
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, 0);
  if (e != hipSuccess) {
    re2e_set_error("re2e_device_ok: %s", hipGetErrorString(e));
    return RE2E_EHIP;
  }
  return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

// ---- stream roles ---------------------------------------------------------------------------------------------------------
// A caller that overlaps bulk work with resident recurrences (JointTrainer's side and weight-gradient streams) marks those
// streams as FILLER streams.  The persistent recurrences keep a workgroup on every CU for a whole sequence, and what they leave
// free of a SIMD's 512 registers (224 beside the forward, 272 beside the backward of the 512-wide layers) admits ONE 4-wave
// engine workgroup (152 registers per wave) but not the 8-wave 256x128 tile (2 waves per SIMD = 304): on a filler stream the
// engine therefore launches the 4-wave tiles (igemm.hip: launch_big).  Everywhere else -- single-stream trainers, benchmarks,
// the latency-critical main stream -- the 8-wave tile stays the default.
namespace {
std::mutex g_role_mu;
hipStream_t g_filler[64];
int g_nfiller = 0;
}  // namespace

extern "C" int re2e_stream_role(hipStream_t stream, int role) {
  RE2E_CHECK_ARG(role == RE2E_STREAM_DEFAULT || role == RE2E_STREAM_FILLER, "role must be RE2E_STREAM_DEFAULT or RE2E_STREAM_FILLER");
  std::lock_guard<std::mutex> lock(g_role_mu);
  int at = -1;
  for (int i = 0; i < g_nfiller; ++i) if (g_filler[i] == stream) at = i;
  if (role == RE2E_STREAM_FILLER) {
    if (at >= 0) return RE2E_OK;
    if (g_nfiller == 64) { re2e_set_error("re2e_stream_role: more than 64 filler streams"); return RE2E_EINVAL; }
    g_filler[g_nfiller++] = stream;
  } else if (at >= 0) {
    g_filler[at] = g_filler[--g_nfiller];
  }
  return RE2E_OK;
}

bool re2e_stream_is_filler(hipStream_t stream) {
  static const bool ignore = exp_env("RE2E_IGNORE_STREAM_ROLE") != nullptr;     // A/B measurements
  if (ignore) return false;
  std::lock_guard<std::mutex> lock(g_role_mu);
  for (int i = 0; i < g_nfiller; ++i) if (g_filler[i] == stream) return true;
  return false;
}
