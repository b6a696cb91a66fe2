// Library-level entry points: version, thread-local error string, device check.
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void re2e_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int re2e_version(void) { return 100; }

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
