// frame_defs.hpp — what frame_post_kernel (frame_kernels.hip) and the env kernel's observation tail
// (frame_tail.hpp) share: the INTER_AREA tap record of the host-built tables and the max-RGB gray arithmetic
// (parl/env/atari_wrappers.py:239, :263-267 as restated in oracle/frame_oracle.c).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace parlhip {
namespace atari {

#ifndef DEVI
#define DEVI __device__ __forceinline__
#endif

struct Tap { int si; float alpha; };   // one INTER_AREA tap: source index, weight (parlhip_frame_post_tables_init)
// The observation tail's view of the same taps (frame_tail.hpp), right after the blob's 32-byte header for the two
// sizes it supports (0 bytes otherwise): x taps lane by lane — Tap[NC][NX][64], column dx = lane + 64 c, tap k; a tap
// past the column's count repeats its first source pixel with weight 0, a column >= dim is (0, 0) — then the M * NY
// y taps of band 0.  One round of independent loads at a fixed address instead of header -> xstart -> taps.
constexpr int kTailHdrBytes = 32;
constexpr int tail_lane_taps_nx(int dim) { return dim == 42 ? 5 : 3; }
constexpr int tail_lane_taps_nc(int dim) { return dim > 64 ? 2 : 1; }
constexpr int tail_lane_taps_bytes(int dim) {
  return (dim == 42 || dim == 84) ? (tail_lane_taps_nc(dim) * tail_lane_taps_nx(dim) * 64 + 8) * 8 : 0;   // (8 Taps of room for the <= 6 y taps)
}

DEVI uint32_t gray_of_colors(uint32_t c0, uint32_t c1, const uint32_t* pal) {
  const uint32_t a = pal[c0 >> 1], b = pal[c1 >> 1];
  const uint32_t r0 = (a >> 16) & 255, g0 = (a >> 8) & 255, b0 = a & 255;
  const uint32_t r1 = (b >> 16) & 255, g1 = (b >> 8) & 255, b1 = b & 255;
  const uint32_t r = r0 > r1 ? r0 : r1, g = g0 > g1 ? g0 : g1, bb = b0 > b1 ? b0 : b1;
  return (r * 4899u + g * 9617u + bb * 1868u + 8192u) >> 14;
}
DEVI uint32_t gray_of_rgb(uint32_t c) {
  return (((c >> 16) & 255) * 4899u + ((c >> 8) & 255) * 9617u + (c & 255) * 1868u + 8192u) >> 14;
}

}  // namespace atari
}  // namespace parlhip
