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

extern "C" int trs_abi_version(void) { return TRS_ABI_VERSION; }

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

// ---- timing events for bench.py (HIP events on the stream the kernels are launched on) --------------------------
extern "C" int trs_events_create(int32_t n, void** handles_out) {
  TRS_REQUIRE(n >= 0 && (n == 0 || handles_out), "trs_events_create: bad arguments");
  for (int32_t i = 0; i < n; ++i) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) {
      trs_set_error("trs_events_create: hipEventCreate failed");
      (void)hipGetLastError();
      return TRS_E_LAUNCH;
    }
    handles_out[i] = (void*)e;
  }
  return TRS_OK;
}

extern "C" int trs_events_destroy(int32_t n, void** handles) {
  for (int32_t i = 0; i < n; ++i)
    if (handles && handles[i]) (void)hipEventDestroy((hipEvent_t)handles[i]);
  return TRS_OK;
}

extern "C" int trs_events_elapsed_ms(void* start, void* stop, float* ms_out) {
  TRS_REQUIRE(start && stop && ms_out, "trs_events_elapsed_ms: NULL argument");
  if (hipEventElapsedTime(ms_out, (hipEvent_t)start, (hipEvent_t)stop) != hipSuccess) {
    trs_set_error("trs_events_elapsed_ms: events not complete (synchronise the stream first)");
    (void)hipGetLastError();
    return TRS_E_LAUNCH;
  }
  return TRS_OK;
}
