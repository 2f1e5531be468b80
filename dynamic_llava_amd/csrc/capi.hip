// ABI plumbing: version, thread-local error string, device check.
#include <stdarg.h>

#include "dl_common.h"

namespace dl {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace dl

extern "C" int dl_version(void) { return DL_ABI_VERSION; }
extern "C" const char* dl_last_error(void) { return dl::g_err; }
extern "C" int dl_device_check(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    dl::set_error("dl_device_check: no HIP device");
    return DL_ERR_ARG;
  }
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) {
    dl::set_error("dl_device_check: hipGetDeviceProperties failed");
    return DL_ERR_ARG;
  }
  if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
    dl::set_error("dl_device_check: device is %s, this library is built for gfx950 only", p.gcnArchName);
    return DL_ERR_ARG;
  }
  return DL_OK;
}
