// common.hpp — shared helpers for libparl_hip.so (gfx950 only; no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/parl_hip.h"

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

// Fixed-order (deterministic) sum of per-workgroup partials partial[part][len] for 16 consecutive
// outputs per 256-thread block: 16 slices of the parts in parallel (slice q takes parts q, q+16, ...,
// four independent accumulators keep four loads in flight), then the 16 slice sums in order through
// LDS.  The result is valid in the threads with (threadIdx.x >> 4) == 0; `red` is 256 floats of LDS.
// The first version of these reductions was one thread per output walking all parts serially:
// 512 parts x 9,264 outputs of the conv12 backward took 119 us (0.16 TB/s), as long as the backward.
__device__ __forceinline__ float partial_sum16(const float* __restrict__ partial, int n_parts, int len, int k,
                                               float* red) {
  const int kk = threadIdx.x & 15, qs = threadIdx.x >> 4;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (k < len) {
    int q = qs;
    for (; q + 48 < n_parts; q += 64) {
      s0 += partial[(size_t)q * len + k];
      s1 += partial[(size_t)(q + 16) * len + k];
      s2 += partial[(size_t)(q + 32) * len + k];
      s3 += partial[(size_t)(q + 48) * len + k];
    }
    for (; q < n_parts; q += 16) s0 += partial[(size_t)q * len + k];
  }
  red[qs * 16 + kk] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  float t = 0.f;
  if (qs == 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i * 16 + kk];
  }
  return t;
}

}  // namespace parlhip
