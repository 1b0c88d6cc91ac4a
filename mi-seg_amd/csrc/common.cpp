// Error reporting + library identity for libmiseg_hip.so
#include "common.h"
#include <string.h>

namespace miseg {
static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace miseg

extern "C" int miseg_abi_version(void) { return 1; }
extern "C" const char* miseg_last_error(void) { return miseg::g_err; }
extern "C" int miseg_device_arch(char* buf, size_t n) {
  if (!buf || n == 0) return MISEG_E_BADARG;
  strncpy(buf, "gfx950", n);
  buf[n - 1] = 0;
  return MISEG_OK;
}
