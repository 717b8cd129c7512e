// frame_tail.hpp — MaxAndSkipEnv max + WarpFrame (frame_post) at the TAIL of an env's two wavefronts
// (atari_env.hip, parlhip_atari_vec_step_obs): the observation leaves the launch that emulated it.
//
// Reference: parl/env/atari_wrappers.py:239 (max over the last two raw frames), :263-267 (WarpFrame:
// gray + cv2.resize INTER_AREA), restated in oracle/frame_oracle.c; arithmetic and float operation order are
// those of frame_post_kernel (frame_kernels.hip), with which this path is bit-identical by test.
//
// Why here: as its own launch frame_post is 27 us (42 in the pipeline) + a launch gap on the actors' critical path,
// and it cannot start before the SLOWEST env of the grid has finished; here every env converts its own pair of
// frames as soon as its picture wave has drawn them, and only the slowest env's conversion is exposed.
// How: the 210 source rows are 42 bands of 5; an output row's INTER_AREA taps never leave its band for dim in
// {42, 84} (210 / 5 = 42 divides dim).  The env's two waves claim the bands in two chunks from a counter in the env's
// ring header (the CPU wave bands 0 .. 17 at its last instruction, the picture wave the rest when it has replayed its
// last records; atari_defs.hpp kObsFirst, atari_core.hpp RenderQueue::obs_next) — no barrier between them, each has
// its own 800-byte gray band in LDS.  Per band: 50 lanes load one uint4 of each colour frame (three bands' loads in
// flight), a band of one colour in both frames is stored as that colour's gray at once; otherwise max + gray through
// the two LDS tables, band to LDS, then lane = output column: the band's y taps times the lane's x taps (registers,
// from the lane-ordered copy at the head of the tables blob).
// LDS: the upper half of the cartridge table's 16 KB (an unbanked 2K cartridge — Pong, Breakout — fills the lower
// half; a 4K cartridge takes the separate launch): no byte added to the env kernel's static LDS, which is what
// lets it sit beside two 70 KB learner workgroups on a CU.
#pragma once
#include <hip/hip_runtime.h>
#include "atari_defs.hpp"
#include "frame_defs.hpp"

namespace parlhip {
namespace atari {

#ifndef DEVI
#define DEVI __device__ __forceinline__
#endif

typedef __attribute__((address_space(3))) uint8_t tail_lds_u8;
typedef __attribute__((address_space(3))) uint32_t tail_lds_u32;
typedef uint32_t tail_u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) tail_u4 tail_lds_u4;
typedef __attribute__((address_space(1))) uint8_t tail_glb_u8;
typedef __attribute__((address_space(1))) const tail_u4 tail_glb_cu4;

#ifndef PARLHIP_TAIL_DEPTH
#define PARLHIP_TAIL_DEPTH 3
#endif
constexpr int kTailBandRows = kObsBandRows;            // source rows per band (5)
constexpr int kTailBands = kH / kTailBandRows;         // 42
constexpr int kTailBandBytes = kTailBandRows * kW;     // 800
constexpr int kTailBandLanes = kTailBandBytes / 16;    // 50 lanes load a band
constexpr int kTailLdsPal = 0, kTailLdsG1 = 512, kTailLdsBands = 1024;   // byte offsets in the free half of rom_lds
constexpr int kTailLdsBytes = kTailLdsBands + 8 * kTailBandBytes;        // 7,424 <= 8,192
static_assert(kH % kTailBandRows == 0 && kTailBandBytes % 16 == 0, "bands are whole uint4s");

__host__ __device__ inline bool obs_tail_supports(int dim, int rom_words) {
  return (dim == 42 || dim == 84) && rom_words == 2048;
}

// the workgroup's colour -> RGB and colour -> gray tables (threads 0 .. 127, before the kernel's first barrier)
DEVI void obs_tail_stage_tables(uint32_t* lds_hi, const uint8_t* blob, int tid) {
  if (tid < 128) {
    const int* hdr = (const int*)blob;
    const uint32_t c = ((const uint32_t*)(blob + hdr[7]) - 128)[tid];
    tail_lds_u8* hi = (tail_lds_u8*)(tail_lds_u32*)lds_hi;
    ((tail_lds_u32*)(hi + kTailLdsPal))[tid] = c;
    hi[kTailLdsG1 + tid] = (uint8_t)gray_of_rgb(c);
  }
}

// Bands [b_begin, b_end) of one env's observation (a claimed chunk or part of it, atari_defs.hpp kObsFirst).  frames: the env's colour frame
// pair; out: its dim x dim slot of the rollout ring; `wave`: which of the workgroup's eight band buffers is this
// wave's.  Arguments arrive in vector registers (a call) and are made wave-uniform again.
// The tap structure of the two supported sizes is fixed (tests/test_frame_oracle_pin.py checks the host-built tables
// for it): dim 42 — one output row per band, its 5 y taps the band's 5 rows, <= 5 x taps per column; dim 84 — two
// output rows per band with 3 y taps each, <= 3 x taps, 84 columns = lanes 0 .. 63 + lanes 0 .. 19.  So every loop
// below has constant bounds: all LDS reads of a band are issued together (as a dynamic tap loop this was one LDS
// round trip per tap: 50 us per env).
template <int DIM>
static __device__ __attribute__((noinline)) void obs_tail_main(const uint8_t* frames_v, uint8_t* out_v, const uint8_t* blob_v,
                                                               uint32_t* lds_hi_v, int b_begin_v, int b_end_v, int single_v, int wave_v) {
  constexpr int M = DIM / kTailBands;             // output rows per band
  constexpr int NY = DIM == 42 ? 5 : 3;           // y taps per output row
  constexpr int NX = DIM == 42 ? 5 : 3;           // x taps per output column (fewer: padded with weight 0)
  constexpr int NC = DIM > 64 ? 2 : 1;            // column sets per lane
  static_assert(DIM == 42 || DIM == 84, "obs_tail_supports");
  auto uni = [](const void* p) -> unsigned long long {
    const unsigned long long x = (unsigned long long)(uintptr_t)p;
    return ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(x >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
  };
  const int lane = (int)(threadIdx.x & 63);
  const int b_begin = __builtin_amdgcn_readfirstlane(b_begin_v), b_end = __builtin_amdgcn_readfirstlane(b_end_v);
  const int single = __builtin_amdgcn_readfirstlane(single_v), wave = __builtin_amdgcn_readfirstlane(wave_v);
  const uint8_t* blob = (const uint8_t*)(uintptr_t)uni(blob_v);
  tail_glb_cu4* f0 = (tail_glb_cu4*)(uintptr_t)uni(frames_v);
  tail_glb_cu4* f1 = single ? f0 : f0 + kFrameBytes / 16;
  tail_glb_u8* out = (tail_glb_u8*)(uintptr_t)uni(out_v);
  tail_lds_u8* hi = (tail_lds_u8*)(tail_lds_u32*)(uint32_t*)(uintptr_t)uni(lds_hi_v);
  const tail_lds_u32* pal = (const tail_lds_u32*)(hi + kTailLdsPal);
  const tail_lds_u8* g1 = hi + kTailLdsG1;
  tail_lds_u8* band = hi + kTailLdsBands + wave * kTailBandBytes;

  // the lane's output columns dx = lane (+ 64 at dim 84) and their x taps, the band's y taps: the lane-ordered copy at
  // the head of the tables (frame_defs.hpp tail_lane_taps_bytes) — independent loads, in flight together with the
  // first bands' pixels (a chunk of three bands is one call of this function: header -> xstart -> taps were three
  // memory round trips before its first pixel)
  // (one 64-bit load per tap, split by shifts: as an int2 vector hipcc 7.2 copied .x over .y after the loads)
  typedef __attribute__((address_space(1))) const unsigned long long glb_tap;   // (si, alpha bits)
  glb_tap* lt = (glb_tap*)(uintptr_t)(blob + kTailHdrBytes);
  static_assert(tail_lane_taps_nx(DIM) == NX && tail_lane_taps_nc(DIM) == NC, "the host's lane-ordered taps");
  unsigned long long xtap[NC][NX];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int k = 0; k < NX; ++k) xtap[c][k] = lt[(c * NX + k) * 64 + lane];
  const unsigned long long ytap = lt[NC * NX * 64 + (lane < M * NY ? lane : 0)];
  // bands b_begin .. b_end - 1: a multiple of kTailDepth of them (the chunks 0 .. 17 / 18 .. 41, or kObsStep * n bands
  // of a chunk claimed from the env's band counter, atari_core.hpp RenderQueue::obs_next)
  const bool loader = lane < kTailBandLanes;
  // the colour pixels of kTailDepth bands are in flight at any time: the frame pair was stored by the picture wave long
  // ago and comes from HBM / the far L2 (1024 envs x 67 KB), one band ahead left a memory round trip per band exposed
  constexpr int kTailDepth = PARLHIP_TAIL_DEPTH;
  static_assert(kObsFirst % kTailDepth == 0 && (kObsBands - kObsFirst) % kTailDepth == 0 && kObsStep % kTailDepth == 0, "the band loop is unrolled by the prefetch depth");
  tail_u4 pa[kTailDepth], pb[kTailDepth];
#pragma unroll
  for (int s = 0; s < kTailDepth; ++s) { pa[s] = tail_u4{0u, 0u, 0u, 0u}; pb[s] = pa[s]; }
#pragma unroll
  for (int s = 0; s < kTailDepth; ++s)
    if (loader) {
      pa[s] = f0[(b_begin + s) * kTailBandLanes + lane];
      pb[s] = f1[(b_begin + s) * kTailBandLanes + lane];
    }
  // (a tap past the column's count carries weight 0 on the column's first source pixel: buf + S * 0 == buf exactly,
  // buf >= 0; the y taps of a band, row by row, are the same (row inside the band, weight) in every band — 210 / dim and
  // dim / 42 are exact in binary — checked on the host-built tables by tests/test_capi_symbols.py)
  int xsi[NC][NX];
  float xal[NC][NX];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int k = 0; k < NX; ++k) { xsi[c][k] = (int)(uint32_t)xtap[c][k]; xal[c][k] = __builtin_bit_cast(float, (uint32_t)(xtap[c][k] >> 32)); }
  int rowoff[M * NY];
  float beta[M * NY];
#pragma unroll
  for (int j = 0; j < M * NY; ++j) {
    rowoff[j] = __builtin_amdgcn_readlane((int)(uint32_t)ytap, j) * kW;
    beta[j] = __builtin_bit_cast(float, __builtin_amdgcn_readlane((int)(uint32_t)(ytap >> 32), j));
  }
  for (int bi0 = b_begin; bi0 < b_end; bi0 += kTailDepth)
#pragma unroll
  for (int s = 0; s < kTailDepth; ++s) {
    const int bi = bi0 + s;
    const tail_u4 a = pa[s], b = pb[s];
    if (loader && bi + kTailDepth < b_end) {
      pa[s] = f0[(bi + kTailDepth) * kTailBandLanes + lane];
      pb[s] = f1[(bi + kTailDepth) * kTailBandLanes + lane];
    }
    // ---- a band of ONE colour in both frames (most of Pong's picture): every output pixel of the band is that colour's
    // gray — the area taps of a constant sum to it within 1e-4 and the result is rounded to the nearest integer
    // (tests/test_frame_oracle_pin.py checks every colour at both sizes on the oracle) — no LDS band, no taps: the
    // env's two waves share one SIMD, and through the observation both are bound by its vector ALU
    {
      const uint32_t c0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.x);
      const bool flat = (((a.x ^ c0) | (a.y ^ c0) | (a.z ^ c0) | (a.w ^ c0) | (b.x ^ c0) | (b.y ^ c0) | (b.z ^ c0) | (b.w ^ c0)) & 0xfefefefeu) == 0u;
      if ((((c0 ^ (c0 >> 8)) & 0x00fefefeu) == 0u) && __ballot(loader && !flat) == 0ull) {
        const uint8_t g = g1[(c0 >> 1) & 127];
#pragma unroll
        for (int r = 0; r < M; ++r)
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            const int dx = lane + 64 * c;
            if (dx < DIM) out[(bi * M + r) * DIM + dx] = g;
          }
        continue;
      }
    }
    // ---- max over the two frames + gray: 16 pixels per lane (frame_post_kernel's fmt == 1 arm)
    {
      const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
      uint32_t ow[4];
      const bool same = (((a.x ^ b.x) | (a.y ^ b.y) | (a.z ^ b.z) | (a.w ^ b.w)) & 0xfefefefeu) == 0u;
      if (__ballot(loader && !same) == 0ull) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          uint32_t o = 0;
#pragma unroll
          for (int j = 0; j < 4; ++j) o |= (uint32_t)g1[(aw[q] >> (8 * j + 1)) & 127] << (8 * j);
          ow[q] = o;
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          uint32_t o = 0;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const uint32_t pa = pal[((aw[q] >> (8 * j)) & 255) >> 1], pb = pal[((bw[q] >> (8 * j)) & 255) >> 1];
            const uint32_t r0 = (pa >> 16) & 255, g0 = (pa >> 8) & 255, b0 = pa & 255;
            const uint32_t r1 = (pb >> 16) & 255, gg1 = (pb >> 8) & 255, b1 = pb & 255;
            const uint32_t r = r0 > r1 ? r0 : r1, g = g0 > gg1 ? g0 : gg1, bb = b0 > b1 ? b0 : b1;
            o |= ((r * 4899u + g * 9617u + bb * 1868u + 8192u) >> 14) << (8 * j);
          }
          ow[q] = o;
        }
      }
      if (loader) ((tail_lds_u4*)band)[lane] = tail_u4{ow[0], ow[1], ow[2], ow[3]};
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- INTER_AREA: lane = output column; float operation order of the oracle (frame_post_kernel):
    // buf = sum_k S[x_k] * alpha_k in tap order, sum = beta_0 * buf_0, then += beta_j * buf_j
    uint8_t px[M][NC][NY][NX];
#pragma unroll
    for (int r = 0; r < M; ++r)
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int j = 0; j < NY; ++j)
#pragma unroll
          for (int k = 0; k < NX; ++k) px[r][c][j][k] = band[rowoff[r * NY + j] + xsi[c][k]];
#pragma unroll
    for (int r = 0; r < M; ++r) {
      const int dy = bi * M + r;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int dx = lane + 64 * c;
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NY; ++j) {
          float buf = 0.f;
#pragma unroll
          for (int k = 0; k < NX; ++k) buf = __fadd_rn(buf, __fmul_rn((float)px[r][c][j][k], xal[c][k]));
          const float tmp = __fmul_rn(beta[r * NY + j], buf);
          sum = (j == 0) ? tmp : __fadd_rn(sum, tmp);
        }
        // cv::saturate_cast<uchar>(float): cvRound (round half to even) then clamp
        int v = (int)__builtin_rintf(sum);
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        if (dx < DIM) out[dy * DIM + dx] = (uint8_t)v;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();   // the band buffer is rewritten by the next iteration
  }
}

// (dim is wave-uniform at the call sites: one branch, two instantiations)
DEVI void obs_tail_dispatch(const uint8_t* frames, uint8_t* out, const uint8_t* blob, uint32_t* lds_hi, int dim, int b_begin,
                            int b_end, int single, int wave) {
  if (dim == 42) obs_tail_main<42>(frames, out, blob, lds_hi, b_begin, b_end, single, wave);
  else obs_tail_main<84>(frames, out, blob, lds_hi, b_begin, b_end, single, wave);
}

}  // namespace atari
}  // namespace parlhip
