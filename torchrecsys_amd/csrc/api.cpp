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

// ---- tuning knobs: the TRS_* environment is read here, once -----------------------------------------------------
#include <mutex>
#include <stdlib.h>

namespace {
struct Knob {
  const char* name;  // without the TRS_ prefix
  int64_t TrsTuning::*wide;
  int TrsTuning::*narrow;
  int64_t unset;
};
const Knob KNOBS[] = {
    {"GRID_CAP", &TrsTuning::grid_cap, nullptr, 256 * 16},
    {"PASS_GRID_CAP", &TrsTuning::pass_grid_cap, nullptr, 4096},
    {"PRESORT_GRID_CAP", &TrsTuning::presort_grid_cap, nullptr, 512},
    {"K1_ITERS", nullptr, &TrsTuning::k1_iters, 0},
    {"PASS_ITERS", nullptr, &TrsTuning::pass_iters, 0},
    {"PASS_NT", nullptr, &TrsTuning::pass_nt, -1},
    {"K1_NT", nullptr, &TrsTuning::k1_nt, -1},
    {"K1_WGS_PER_CU", nullptr, &TrsTuning::k1_wgs_per_cu, 2},
    {"GEMM32_NO_GLDS", nullptr, &TrsTuning::gemm32_no_glds, 0},
    {"GEMM16_TN_WIDE", nullptr, &TrsTuning::gemm16_tn_wide, -1},
    {"GEMM16_TILE", nullptr, &TrsTuning::gemm16_tile, 0},
    {"GEMM16_NO_GLDS", nullptr, &TrsTuning::gemm16_no_glds, 0},
    {"BN_FINAL_TWO_SWEEPS", nullptr, &TrsTuning::bn_final_two_sweeps, 0},
};
TrsTuning g_tuning;
std::once_flag g_tuning_once;
void knob_store(const Knob& k, int64_t v) {
  if (k.wide) g_tuning.*(k.wide) = v;
  else g_tuning.*(k.narrow) = (int)v;
}
}  // namespace

TrsTuning& trs_tuning() {
  std::call_once(g_tuning_once, [] {
    for (const Knob& k : KNOBS) {
      char name[64];
      snprintf(name, sizeof(name), "TRS_%s", k.name);
      const char* e = getenv(name);
      knob_store(k, (e && *e) ? atoll(e) : k.unset);
    }
  });
  return g_tuning;
}

// name: a knob of TrsTuning without the TRS_ prefix ("GEMM16_TILE"); unset != 0 restores the knob's default.
extern "C" int trs_tuning_set(const char* name, int64_t value, int32_t unset) {
  TRS_REQUIRE(name != nullptr, "trs_tuning_set: NULL name");
  (void)trs_tuning();
  for (const Knob& k : KNOBS)
    if (strcmp(k.name, name) == 0) {
      knob_store(k, unset ? k.unset : value);
      return TRS_OK;
    }
  trs_set_error("trs_tuning_set: unknown knob %s", name);
  return TRS_E_ARG;
}

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
