// philox.hpp — Philox4x32-10 (Salmon et al., SC'11) counter-based RNG shared by the sampling
// kernel and the env reset logic.  key = seed; counter = (offset, row).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace parlhip {

__device__ __forceinline__ void philox4x32_10(uint64_t seed, uint64_t offset, uint64_t row,
                                              uint32_t out[4]) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  uint32_t c0 = (uint32_t)offset, c1 = (uint32_t)(offset >> 32), c2 = (uint32_t)row,
           c3 = (uint32_t)(row >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ double philox_uniform53(uint64_t seed, uint64_t offset, uint64_t row) {
  uint32_t w[4];
  philox4x32_10(seed, offset, row, w);
  const uint64_t a = w[0] >> 5, b = w[1] >> 6;  // numpy random_sample construction
  return (double)(a * 67108864ull + b) / 9007199254740992.0;
}

}  // namespace parlhip
