// common.hpp — shared helpers for libparl_hip.so (gfx950 only; no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../include/parl_hip.h"

namespace parlhip {

extern thread_local int g_last_hip_error;

inline int check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    g_last_hip_error = (int)e;
    return PARLHIP_ELAUNCH;
  }
  return PARLHIP_OK;
}

inline int check(hipError_t e) {
  if (e != hipSuccess) {
    g_last_hip_error = (int)e;
    return PARLHIP_ELAUNCH;
  }
  return PARLHIP_OK;
}

#define PARLHIP_EXPORT extern "C" __attribute__((visibility("default")))

constexpr int kWave = 64;          // CDNA4 wavefront
constexpr int kNumCU = 256;        // MI355X
constexpr int kNumXCD = 8;

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace parlhip
