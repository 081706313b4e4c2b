// C ABI plumbing: error reporting, layout helpers.
#include <stdarg.h>
#include <stdio.h>

#include "sd_common.h"

namespace sda {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return -2;
  }
  return 0;
}

}  // namespace sda

extern "C" int sda_abi_version(void) { return SDA_ABI_VERSION; }
extern "C" const char* sda_last_error(void) { return sda::g_err; }
extern "C" long sda_rows_alloc(int B, int T) { return sda::rows_alloc(B, T); }
extern "C" int sda_pad_channels(int C) { return (C + SDA_CH_ALIGN - 1) / SDA_CH_ALIGN * SDA_CH_ALIGN; }
extern "C" int sda_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
