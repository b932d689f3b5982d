// api.cpp — error plumbing and device check of libtrs_hip.so.
#include "trs_common.h"

#include <string.h>

static thread_local char g_err[512] = "";

void trs_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* trs_last_error(void) { return g_err; }

extern "C" int trs_abi_version(void) { return 1; }

extern "C" int trs_check_device(void) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) {
    trs_set_error("trs_check_device: no HIP device");
    (void)hipGetLastError();
    return TRS_E_DEVICE;
  }
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) {
    trs_set_error("trs_check_device: hipGetDeviceProperties failed");
    (void)hipGetLastError();
    return TRS_E_DEVICE;
  }
  if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
    trs_set_error("trs_check_device: device %d is %s, this library is built for gfx950 only", dev, p.gcnArchName);
    return TRS_E_DEVICE;
  }
  return TRS_OK;
}
