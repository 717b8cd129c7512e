// conv_kernels.hip — fused conv1 + conv2 of the IMPALA Atari network on gfx950 MFMA, reading
// the uint8 observations of the rollout ring directly.
//
// Reference: examples/IMPALA/atari_model.py:21-90 (AtariModel): obs / 255 (:66),
// conv1 4->16 k4 s2 p1 + ReLU (42x42 -> 21x21), conv2 16->32 k4 s2 p2 + ReLU (-> 11x11); in the
// reference every actor runs these as cuDNN/CPU convs on a batch of 5 per env step
// (examples/IMPALA/atari_agent.py:25-42).  Here the actor path runs them for all envs of a GPU
// in ONE kernel per env step: no im2col matrices in HBM (the GEMM-lowered convs write and re-read
// 115 + 127 MB per step at 1024 envs), no intermediate activations in HBM.
//
// One workgroup (4 wavefronts) per observation.  Both convolutions are implicit GEMMs on the
// f32 matrix cores (v_mfma_f32_16x16x4_f32: exact f32 FMA chains), weights held in registers:
//   conv1: [441 positions x 64] x [64 x 16]   A gathered from the zero-padded input in LDS
//   conv2: [121 positions x 256] x [256 x 32] A gathered from the zero-padded conv1 output in LDS
// with k = c*16 + kh*4 + kw (the order of weight.flatten(1)).  Operand maps (guide: A[l&15][l>>4],
// B[l>>4][l&15], D col = l&15, row = 4*(l>>4) + reg).
#include "common.hpp"

namespace parlhip {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef PARLHIP_CONV_REGIONS  // diagnostic build only (tools/conv_regions.py): s_memtime clocks per phase, summed over
__device__ unsigned long long g_conv_regions[16];   // the workgroups' wave 0; [15] = observations
#define CONV_REGION(i)                                                                       \
  do {                                                                                       \
    const unsigned long long t_ = __builtin_readcyclecounter();                              \
    if (threadIdx.x == 0) atomicAdd(&g_conv_regions[i], t_ - creg_t);                         \
    creg_t = t_;                                                                             \
  } while (0)
#define CONV_REGION_BEGIN() unsigned long long creg_t = __builtin_readcyclecounter()
#else
#define CONV_REGION(i) do {} while (0)
#define CONV_REGION_BEGIN() do {} while (0)
#endif

// The LDS reads above this line are ISSUED above it (the compiler may not move memory operations across): with the
// register file nearly full the scheduler otherwise places every read right in front of the MFMA that waits for it.
#define LDS_FENCE() asm volatile("" ::: "memory")

// Fill phases (round 4).  Every kernel here copies its observation's inputs from HBM into LDS before its MFMA
// phases.  Written as `for (i = tid; i < N; i += 256) lds[f(i)] = g(src[i])` the compiler kept ONE load in flight per
// iteration (`global_load; s_waitcnt vmcnt(0); ds_write` in the ISA of rounds 1-3): N / 256 HBM round trips per
// observation, one wave per SIMD and nobody to hide them — for conv1_84_bwd 40 round trips = 30 of the 42 us it
// spent per observation.  A Batch issues ALL loads of a thread first (COUNT / 256 registers) and consumes them
// afterwards; where the register file allows, the loads of the workgroup's NEXT observation are issued before the
// MFMA phases of the current one and consumed after them (the `pre` objects below).
template <int COUNT, typename T>
struct Batch {
  static constexpr int kPer = (COUNT + 255) / 256;
  T v[kPer];
  __device__ __forceinline__ void load(const T* __restrict__ src, int tid) {
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const int i = tid + 256 * j;
      if ((j + 1) * 256 <= COUNT || i < COUNT) v[j] = src[i];
    }
  }
  template <typename L>
  __device__ __forceinline__ void load_by(int tid, L&& address_of) {   // address_of(i) -> const T*
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const int i = tid + 256 * j;
      if ((j + 1) * 256 <= COUNT || i < COUNT) v[j] = *address_of(i);
    }
  }
  template <typename F>
  __device__ __forceinline__ void each(int tid, F&& f) const {
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const int i = tid + 256 * j;
      if ((j + 1) * 256 <= COUNT || i < COUNT) f(i, v[j]);
    }
  }
};

// 16 bytes per lane from global memory straight into LDS (global_load_lds_dwordx4): the wave's 1 KiB lands at the
// wave-uniform `lds_chunk` + 16 * lane, no staging registers, no ds_write; completion is counted on vmcnt (the fence of
// the next __syncthreads() waits for it).
__device__ __forceinline__ void glds16(const float* gsrc_lane, float* lds_chunk) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                   (__attribute__((address_space(3))) void*)lds_chunk, 16, 0, 0);
}

// (float)u / 255.0f without the division: one Newton step makes the product correctly rounded for u = 0 .. 255
__device__ __forceinline__ float byte_over_255(uint32_t u) {
  const float x = (float)u, r = 1.0f / 255.0f;
  const float q = x * r;
  return __builtin_fmaf(__builtin_fmaf(-q, 255.0f, x), r, q);
}

constexpr int kD = 42, kP1 = 44;          // input, zero-padded input (pad 1, +1 slack column/row)
constexpr int kO1 = 21, kC1 = 16;         // conv1 output size / channels
constexpr int kP2 = 25;                   // zero-padded conv1 output (pad 2)
constexpr int kO2 = 11, kC2 = 32;         // conv2 output size / channels
constexpr int kM2 = kO2 * kO2;
constexpr int kK1 = 64, kK2 = 256;
constexpr int kLdsIn = 4 * kP1 * kP1;     // 7744 floats
constexpr int kLdsC1 = kC1 * kP2 * kP2;   // 10000
constexpr int kLdsFloats = kLdsIn + kLdsC1;  // 17,744 floats = 70,976 B: two workgroups per CU
constexpr int kA1Row = kLdsC1;               // floats per observation of a saved conv1 activation (the padded tile)

// Both weight matrices live in registers (B operands: 16 + 128 VGPRs per lane, loaded once per
// workgroup), so an MFMA needs one LDS gather (conv1) or half of one (conv2: both N-tiles reuse
// the A value); outputs leave the accumulators straight for HBM; the zero borders of the two LDS
// tiles are written once and never touched again (only interiors are rewritten per observation).
// RING: the observation of env n is not a materialised [4,42,42] stack but the four frames of the rollout ring it
// consists of (FrameStack as the ring keeps it: frame j of the stack at `slot` is the frame `min(3 - j, since)` slots
// back, `since` = the env's steps since its last reset, clamped at 3 — stack_gather_kernel's rule): the actors' step
// reads them in place and the gather launch (11 us + a launch gap per env step) goes away.
struct RingObs {
  const uint8_t* ring;     // [num_slots][E][d*d] (d = 42: conv12_u8_mfma_kernel, d = 84: conv1_84_u8_mfma_kernel)
  const uint8_t* since;    // [num_slots][E]
  int num_slots, E, slot;
};

// Operand-order copy of the two weight matrices (parlhip_atari42_conv12_weights_f32): register r of lane
// (q = lane >> 4, col = lane & 15) is bw1[r] for r < 16 and bw2[(r - 16) >> 1][(r - 16) & 1] after it; four registers
// per float4, [36][64] float4s (the FORWARD region) — a wave reads its 144 operands with 36 fully coalesced 1 KB loads.  From the
// nn.Conv2d layout the same operands are 144 dword loads that touch 16 cache lines each (sixteen weight rows 1 KB
// apart): measured as THE start-up cost of a workgroup — 29 of the 67 us of the actors' 1024-observation launch
// (tools/conv12_scaling.py with -DPARLHIP_CONV12_ABL=1: 20.0 / 34.6 us for nothing but the weight fetch at 256 /
// 512 workgroups), the vector L1's tag rate, not bytes.
// A second region behind it serves the backward kernel (conv12_bwd_u8_mfma_kernel, step (3)): for wave w (parity class
// py = w >> 1, px = w & 1) and lane (q, col) (ta = q >> 1, tb = q & 1) register o of bt[32] is
// w2[o][col][py + 2 ta][px + 2 tb]: [4 waves][8][64] float4s, every element of w2 exactly once.  (Its bw1 is the
// forward region's first 16 registers.)
constexpr int kPackedRegs = 16 + 128, kPackedFwdFloats = kPackedRegs * 64;   // 9,216 floats
constexpr int kPackedBwdFloats = 4 * 32 * 64;                                // 8,192 floats
constexpr int kPackedFloats = kPackedFwdFloats + kPackedBwdFloats;           // 17,408 floats = 69,632 B
__global__ __launch_bounds__(256) void conv12_weights_pack_kernel(const float* __restrict__ w1, const float* __restrict__ w2,
                                                                  float* __restrict__ packed) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= kPackedFloats) return;
  if (i >= kPackedFwdFloats) {
    const int j = i - kPackedFwdFloats;
    const int w = j >> 11, o = (j >> 6) & 31, lane = j & 63, q = lane >> 4, col = lane & 15;
    const int py = w >> 1, px = w & 1, ta = q >> 1, tb = q & 1;
    packed[kPackedFwdFloats + ((((w * 8 + (o >> 2)) * 64 + lane) << 2) + (o & 3))] =
        w2[o * kK2 + col * 16 + (py + 2 * ta) * 4 + (px + 2 * tb)];
    return;
  }
  const int r = i >> 6, lane = i & 63, q = lane >> 4, col = lane & 15;
  float v;
  if (r < 16) {
    v = w1[col * kK1 + r * 4 + q];
  } else {
    const int rr = r - 16, ks = rr >> 1, t = rr & 1;
    v = w2[(16 * t + col) * kK2 + ks * 4 + q];
  }
  packed[(((r >> 2) * 64 + lane) << 2) + (r & 3)] = v;
}

// PACKED: `packed` holds the operand-order weights (w1 / w2 unused); otherwise w1 / w2 in the nn.Conv2d layout.  Two
// instantiations, not a run-time branch: with both fetch sequences in one function the allocator went from 240 to 280
// registers (24 of them AGPRs) — one wave per SIMD instead of two, 67 -> 78 us per 1024 observations.
// SAVE (round 6, the learner's forward): the conv1 activation leaves for HBM as the zero-padded LDS tile it is
// ([16][25][25] floats, kA1Row per observation, 16-byte copies, no index arithmetic) and the backward kernel loads it
// instead of recomputing conv1 (conv12_bwd_u8_mfma_kernel<.., true>).
template <bool RING, bool PACKED, bool SAVE>
__global__ __launch_bounds__(256, 2) void conv12_u8_mfma_kernel(
    const uint8_t* __restrict__ obs, RingObs ro, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ out, int n_obs,
    const float* __restrict__ packed, float* __restrict__ a1_out) {
  extern __shared__ float lds[];
  float* in_pad = lds;                  // [4][44][44]
  float* c1_pad = in_pad + kLdsIn;      // [16][25][25]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, col = lane & 15;
  // B[k][n] = w[n][k]: lane (q, col) holds k = 4*ks + q, n = col (+16 for the second N-tile)
  float bw1[16], bw2[64][2];
  if constexpr (PACKED) {
    const float4* pk = reinterpret_cast<const float4*>(packed) + lane;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 v = pk[g * 64];
      bw1[4 * g] = v.x; bw1[4 * g + 1] = v.y; bw1[4 * g + 2] = v.z; bw1[4 * g + 3] = v.w;
    }
#pragma unroll
    for (int g = 0; g < 32; ++g) {
      const float4 v = pk[(4 + g) * 64];
      bw2[2 * g][0] = v.x; bw2[2 * g][1] = v.y; bw2[2 * g + 1][0] = v.z; bw2[2 * g + 1][1] = v.w;
    }
  } else {
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) bw1[ks] = w1[col * kK1 + ks * 4 + q];
#pragma unroll
    for (int ks = 0; ks < 64; ++ks) {
      bw2[ks][0] = w2[col * kK2 + ks * 4 + q];
      bw2[ks][1] = w2[(16 + col) * kK2 + ks * 4 + q];
    }
  }
  const float bias1 = b1[col], bias20 = b2[col], bias21 = b2[16 + col];
#ifdef PARLHIP_CONV12_ABL   // diagnostic builds (tools/build_obj_variant.sh c12aN conv_kernels.hip -DPARLHIP_CONV12_ABL=N): where a workgroup's start goes
#define C12_ABL_EXIT(N) if (PARLHIP_CONV12_ABL == N) { float z = bias1 + bias20 + bias21; for (int ks = 0; ks < 16; ++ks) z += bw1[ks]; for (int ks = 0; ks < 64; ++ks) z += bw2[ks][0] + bw2[ks][1]; if (z == 123.456f) out[tid] = z + lds[tid]; return; }
#else
#define C12_ABL_EXIT(N)
#endif
  C12_ABL_EXIT(1)
  // conv1: position m = 16 wave + col (gathers) and 16 wave + 4 q (first D row) of this wave's first tile
  const int oy1 = (16 * wave + col) / kO1, ox1 = (16 * wave + col) - oy1 * kO1;
  const int ey1 = (16 * wave + 4 * q) / kO1, ex1 = (16 * wave + 4 * q) - ey1 * kO1;
  for (int i = tid; i < kLdsFloats; i += 256) lds[i] = 0.0f;
  const bool words = RING || (reinterpret_cast<uintptr_t>(obs) & 3) == 0;   // 7056 B per observation: true for all or none
  Batch<kD * kD, uint32_t> pre;                                      // 4 * 1764 bytes = 1764 words: 7 per thread
  auto fetch = [&](int n) {
    if constexpr (RING) {
      const int sr = ro.since[(size_t)ro.slot * ro.E + n];           // wave-uniform
      const uint32_t* fr[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int from = ro.slot - (3 - j < sr ? 3 - j : sr);
        from += from < 0 ? ro.num_slots : 0;
        fr[j] = reinterpret_cast<const uint32_t*>(ro.ring + ((size_t)from * ro.E + n) * (kD * kD));   // 1764 B: a word multiple
      }
      pre.load_by(tid, [&](int wi) {
        const int c = wi / (kD * kD / 4);
        const uint32_t* f = c == 0 ? fr[0] : (c == 1 ? fr[1] : (c == 2 ? fr[2] : fr[3]));
        return f + (wi - c * (kD * kD / 4));
      });
    } else {
      pre.load(reinterpret_cast<const uint32_t*>(obs + (size_t)n * 4 * kD * kD), tid);
    }
  };
  if (words && (int)blockIdx.x < n_obs) fetch(blockIdx.x);
  for (int n = blockIdx.x; n < n_obs; n += gridDim.x) {
    __syncthreads();  // borders zeroed / the previous observation's conv2 gathers are done
    C12_ABL_EXIT(2)
    // ---- obs u8 -> padded float input (x / 255, the division as in the reference) ----
    const uint8_t* src = obs + (size_t)n * 4 * kD * kD;
    if (words) {  // wave-uniform
      pre.each(tid, [&](int wi, uint32_t v) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = wi * 4 + j;
          const int c = i / (kD * kD), r = i - c * kD * kD, y = r / kD, x = r - y * kD;
          in_pad[c * kP1 * kP1 + (y + 1) * kP1 + (x + 1)] = (float)((v >> (8 * j)) & 255u) / 255.0f;
        }
      });
      if (n + (int)gridDim.x < n_obs) fetch(n + gridDim.x);   // the next observation's bytes are in flight during this one's MFMAs
    } else {
      for (int i = tid; i < 4 * kD * kD; i += 256) {
        const int c = i / (kD * kD), r = i - c * kD * kD, y = r / kD, x = r - y * kD;
        in_pad[c * kP1 * kP1 + (y + 1) * kP1 + (x + 1)] = (float)src[i] / 255.0f;
      }
    }
    __syncthreads();
    C12_ABL_EXIT(3)
    // ---- conv1: 28 M-tiles of 16 positions, 7 per wave ----
    // Positions advance incrementally (a tile is 64 positions further = 3 rows + 1 column of 21; rounds 1-3 divided
    // by 21 five times per tile).  Round 6: TWO tiles per step (t and t + 4) on two accumulators — a tile's 16 MFMAs
    // are one dependent chain (40 clocks per MFMA instead of 32, and nothing to issue while its operands are on
    // their way): with a second, independent chain interleaved the pipe sees a back-to-back stream and one LDS wait
    // per 32 MFMAs.  The order of a tile's sum is unchanged (bit-identical a1).
    {
      int oy = oy1, ox = ox1, ey = ey1, ex = ex1;
#pragma unroll 1
      for (int t = wave; t < 28; t += 8) {
        const bool two = t + 4 < 28;                          // wave-uniform (the last step of a wave: one tile)
        int oyb = oy + 3, oxb = ox + 1;                       // tile t + 4: 64 positions further
        if (oxb >= kO1) { oxb -= kO1; oyb += 1; }
        int eyb = ey + 3, exb = ex + 1;
        if (exb >= kO1) { exb -= kO1; eyb += 1; }
        const bool in = oy < kO1, inb = oyb < kO1;            // m < 441 (the last tile: clamp to the last position)
        const float* a_base = in_pad + (2 * (in ? oy : kO1 - 1)) * kP1 + 2 * (in ? ox : kO1 - 1) + q;
        const float* b_base = in_pad + (2 * (inb ? oyb : kO1 - 1)) * kP1 + 2 * (inb ? oxb : kO1 - 1) + q;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, accb = {0.f, 0.f, 0.f, 0.f};
        if (two) {
          float av[16], bv[16];
#pragma unroll
          for (int ks = 0; ks < 16; ++ks) {
            av[ks] = a_base[(ks >> 2) * kP1 * kP1 + (ks & 3) * kP1];
            bv[ks] = b_base[(ks >> 2) * kP1 * kP1 + (ks & 3) * kP1];
          }
          LDS_FENCE();
#pragma unroll
          for (int ks = 0; ks < 16; ++ks) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], bw1[ks], acc, 0, 0, 0);
            accb = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[ks], bw1[ks], accb, 0, 0, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);   // (ds_read2: two operands per instruction)
          __builtin_amdgcn_sched_group_barrier(0x008, 32, 0);
        } else {
#pragma unroll
          for (int ks = 0; ks < 16; ++ks) {
            const float a = a_base[(ks >> 2) * kP1 * kP1 + (ks & 3) * kP1];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw1[ks], acc, 0, 0, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        }
        auto put = [&](const f32x4& d, int ey_, int ex_) {   // rows 4q + r of D: positions (ey, ex + r), wrapping into the next row
          float* crow = c1_pad + col * kP2 * kP2 + (ey_ + 2) * kP2 + (ex_ + 2);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool w = ex_ + r >= kO1;
            const int yy = ey_ + (w ? 1 : 0);
            if (yy < kO1) {
              const float v = d[r] + bias1;
              crow[r + (w ? kP2 - kO1 : 0)] = v > 0.f ? v : 0.f;
            }
          }
        };
        put(acc, ey, ex);
        if (two) put(accb, eyb, exb);
        oy += 6; ox += 2;                                     // 128 positions further = 6 rows + 2 columns
        if (ox >= kO1) { ox -= kO1; oy += 1; }
        ey += 6; ex += 2;
        if (ex >= kO1) { ex -= kO1; ey += 1; }
      }
    }
    __syncthreads();
    C12_ABL_EXIT(4)
    if constexpr (SAVE) {   // the padded conv1 tile as it stands in LDS (borders included; the last 16 floats are border)
      float4* d4 = reinterpret_cast<float4*>(a1_out + (size_t)n * kA1Row);
      const float4* s4 = reinterpret_cast<const float4*>(c1_pad);
#pragma unroll 2
      for (int i = tid; i < kLdsC1 / 4; i += 256) d4[i] = s4[i];
    }
    // ---- conv2: 8 M-tiles, 2 per wave, both N-tiles per A gather; D goes straight to HBM ----
    float* dst = out + (size_t)n * kC2 * kM2;
    for (int mt = wave; mt < 8; mt += 4) {
      int m = mt * 16 + col;
      m = m < kM2 ? m : kM2 - 1;
      const int oy = m / kO2, ox = m - oy * kO2;
      const float* a_base = c1_pad + (2 * oy) * kP2 + 2 * ox + q;
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 64; ++ks) {
#if defined(PARLHIP_CONV12_ABL) && PARLHIP_CONV12_ABL == 6    // (diagnostic: conv2's MFMAs without its LDS gathers)
        const float a = bias20 + (float)ks;
#else
        const float a = a_base[(ks >> 2) * kP2 * kP2 + (ks & 3) * kP2];
#endif
#if defined(PARLHIP_CONV12_ABL) && PARLHIP_CONV12_ABL == 7    // (diagnostic: conv2's LDS gathers without its MFMAs)
        acc0[0] += a * bw2[ks][0];
        acc1[0] += a * bw2[ks][1];
#else
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw2[ks][0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw2[ks][1], acc1, 0, 0, 0);
#endif
      }
      // the order of the block for the scheduler (ds_read2: two operands per instruction): 8 operands, then seven times
      // [the next 8 | the 16 MFMAs of the 8 before them], then the last 16 MFMAs — 16 operand registers in flight
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
      for (int g = 0; g < 7; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
      // D: column = channel, rows 4q..4q+3 = 4 consecutive positions of the [32][121] row
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mo = mt * 16 + q * 4 + r;
#ifdef PARLHIP_CONV12_ABL
        if (PARLHIP_CONV12_ABL == 5 && acc0[r] != 123.456f) continue;   // (diagnostic: everything but the output stores)
#endif
        if (mo < kM2) {
          const float v0 = acc0[r] + bias20, v1 = acc1[r] + bias21;
          dst[col * kM2 + mo] = v0 > 0.f ? v0 : 0.f;
          dst[(16 + col) * kM2 + mo] = v1 > 0.f ? v1 : 0.f;
        }
      }
    }
  }
}


// ----------------------------------------------------------------------------------------
// Backward of the fused conv1 + conv2 for the LEARNER (IMPALA.learn, impala.py:148-215 runs
// AtariModel.policy / .value under autograd; the gradient of the two convolutions w.r.t. their
// weights and biases is what this kernel produces — the observations need no gradient).
//
// Inputs per observation: the uint8 stack (7,056 B), the forward output a2 = relu(conv2) and the
// incoming gradient dY w.r.t. a2 (15,488 B each).  Nothing else crosses HBM: conv1 is recomputed
// into LDS exactly as the forward kernel computes it (same operand order, bit-identical), so the
// 28 KB / observation conv1 activation is never stored (1.44 GB per 51,200-observation update),
// and no im2col / col2im matrix exists (the GEMM-lowered convolutions moved ~30 GB per update and
// spent 19 ms in two split-K-less dW GEMMs with 16x64 / 32x256 outputs).
//
// One workgroup (4 waves) per observation, grid-stride; four implicit GEMMs on the f32 matrix
// cores per observation, gradients accumulated in registers over all observations of the workgroup:
//   (1) conv1 forward (recompute)                [441 x 64] x [64 x 16]
//   (2) dW2[o][k] += sum_p dz2[o][p] patch2[p][k]   [32 x 121] x [121 x 256], dz2 = dY * (a2 > 0)
//   (3) dA1 = transposed conv of dz2 with w2, as a GATHER per output parity class (stride 2,
//       kernel 4: a conv1 output (y, x) receives from the 2 x 2 conv2 outputs (iy+1-a, ix+1-b)
//       through taps (py+2a, px+2b)); one class per wave: [<=121 x 128] x [128 x 16];
//       dz1 = dA1 * (a1 > 0) overwrites a1 in place in LDS
//   (4) dW1[c][k] += sum_p dz1[c][p] patch1[p][k]   [16 x 441] x [441 x 64]
// Bias gradients are the row sums of dz2 / dz1, accumulated per lane on the way.  Every workgroup
// writes its partial sums; conv12_bwd_reduce_kernel adds them in a fixed order (deterministic).
// ----------------------------------------------------------------------------------------
constexpr int kZ2W = 13, kZ2H = 12, kZ2 = kZ2H * kZ2W;   // dz2, zero-padded: row 11 / columns 11-12 stay zero
constexpr int kBwdDW1 = 0, kBwdDB1 = 1024, kBwdDW2 = 1040, kBwdDB2 = 1040 + 8192;
constexpr int kBwdPartial = 1040 + 8192 + 32;            // 9264 floats per workgroup
// LDS: the observation stays uint8 (zero-padded [4][44][44] = 7,744 B) next to a 256-entry table of
// (float)u / 255.0f — the forward kernel's exact operand values — so the workgroup needs 69 KB instead
// of 92: two workgroups per CU, and the actors' conv12 kernel (71 KB) can share the CU.
constexpr int kLdsBwdU8 = 4 * kP1 * kP1;                                   // 7,744 bytes = 1,936 floats
constexpr int kLdsBwdFloats = kLdsBwdU8 / 4 + 256 + kLdsC1 + 32 * kZ2 + 384;  // 17,568 floats = 70,272 B

// PACKED: `w1` is parlhip_atari42_conv12_weights_f32's buffer (operand order), w2 unused.
// HAVE_A1 (round 6): `a1` holds the forward kernel's conv1 activation as padded tiles (conv12_u8_mfma_kernel<.., SAVE>,
// kA1Row floats per observation) and phase (1) is a 40 KB copy instead of 448 MFMAs behind two dependent LDS gathers
// per operand — per-phase clocks of round 4 (profiles/r04_conv12_bwd_regions.log): conv1 15.9 k of 72.5 k per observation
// and wave at 142 clocks per MFMA.  At the learner's 1000-row updates the tiles are 40 MB written and read once per
// update; the recompute stays for callers without the buffer (and for batches where 40 KB per row would not pay).
template <bool PACKED, bool HAVE_A1>
__global__ __launch_bounds__(256, 2) void conv12_bwd_u8_mfma_kernel(
    const uint8_t* __restrict__ obs, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ a2, const float* __restrict__ dy,
    float* __restrict__ partial, int n_obs, const float* __restrict__ a1) {
  extern __shared__ float lds[];
  uint8_t* in_u8 = reinterpret_cast<uint8_t*>(lds);   // [4][44][44] uint8, zero-padded
  float* lut = lds + kLdsBwdU8 / 4;                   // [256] (float)u / 255.0f
  float* c1_pad = lut + 256;                          // [16][25][25]: a1, then dz1 in place
  float* dz2p = c1_pad + kLdsC1;                      // [32][12][13]
  float* red = dz2p + 32 * kZ2;                       // [384] bias-gradient staging
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, col = lane & 15;
  float bw1[16];
  const float bias1 = b1[col];
  // (3): this wave's parity class and its B operand  B[k = (o, a, b)][n = c] = w2[o][c][py+2a][px+2b]
  const int py = wave >> 1, px = wave & 1, ta = q >> 1, tb = q & 1;
  float bt[32];
  if constexpr (PACKED) {   // 4 + 8 coalesced float4 loads instead of 48 dword loads over 16 cache lines each
    if constexpr (!HAVE_A1) {
      const float4* pk = reinterpret_cast<const float4*>(w1) + lane;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 v = pk[g * 64];
        bw1[4 * g] = v.x; bw1[4 * g + 1] = v.y; bw1[4 * g + 2] = v.z; bw1[4 * g + 3] = v.w;
      }
    }
    const float4* pb = reinterpret_cast<const float4*>(w1 + kPackedFwdFloats) + wave * 8 * 64 + lane;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const float4 v = pb[g * 64];
      bt[4 * g] = v.x; bt[4 * g + 1] = v.y; bt[4 * g + 2] = v.z; bt[4 * g + 3] = v.w;
    }
  } else {
    if constexpr (!HAVE_A1) {
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) bw1[ks] = w1[col * kK1 + ks * 4 + q];
    }
#pragma unroll
    for (int o = 0; o < 32; ++o) bt[o] = w2[o * kK2 + col * 16 + (py + 2 * ta) * 4 + (px + 2 * tb)];
  }
  f32x4 acc2[2][4], acc1 = {0.f, 0.f, 0.f, 0.f}, acc1b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc2[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float db2a0 = 0.f, db2a1 = 0.f, db1a = 0.f;
  for (int i = tid; i < kLdsBwdFloats; i += 256) lds[i] = 0.0f;
  __syncthreads();
  lut[tid] = (float)tid / 255.0f;
  const int kh = col >> 2, kw = col & 3;   // tap of this lane's k column in (2) and (4)
  // (1): position m = 16 wave + col (gathers) and 16 wave + 4 q (first D row) of this wave's first tile; (3): m = col, 4 q
  const int oy1 = (16 * wave + col) / kO1, ox1 = (16 * wave + col) - oy1 * kO1;
  const int ey1 = (16 * wave + 4 * q) / kO1, ex1 = (16 * wave + 4 * q) - ey1 * kO1;
  const int iy3 = col / (kO2 - px), ix3 = col - iy3 * (kO2 - px);
  const int jy3 = (4 * q) / (kO2 - px), jx3 = 4 * q - jy3 * (kO2 - px);
  const bool words = (reinterpret_cast<uintptr_t>(obs) & 3) == 0;   // 7056 B per observation: true for all or none
  Batch<kD * kD, uint32_t> pre_x;                                    // the NEXT observation's inputs (see Batch)
  Batch<kC2 * kM2, float> pre_a, pre_d;
  if ((int)blockIdx.x < n_obs) {
    if (words) pre_x.load(reinterpret_cast<const uint32_t*>(obs + (size_t)blockIdx.x * 4 * kD * kD), tid);
    pre_a.load(a2 + (size_t)blockIdx.x * kC2 * kM2, tid);
    pre_d.load(dy + (size_t)blockIdx.x * kC2 * kM2, tid);
  }
  CONV_REGION_BEGIN();
#pragma unroll 1
  for (int n = blockIdx.x; n < n_obs; n += gridDim.x) {
    __syncthreads();
    CONV_REGION(0);   // waiting for the other waves (end of the previous observation)
    // ---- obs u8 -> zero-padded u8 tile ----
    const uint8_t* src = obs + (size_t)n * 4 * kD * kD;
    if (words) {
      pre_x.each(tid, [&](int wi, uint32_t v) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = wi * 4 + j;
          const int c = i / (kD * kD), r = i - c * kD * kD, y = r / kD, x = r - y * kD;
          in_u8[c * kP1 * kP1 + (y + 1) * kP1 + (x + 1)] = (uint8_t)(v >> (8 * j));
        }
      });
    } else {
      for (int i = tid; i < 4 * kD * kD; i += 256) {
        const int c = i / (kD * kD), r = i - c * kD * kD, y = r / kD, x = r - y * kD;
        in_u8[c * kP1 * kP1 + (y + 1) * kP1 + (x + 1)] = src[i];
      }
    }
    // ---- dz2 = dY * (a2 > 0) -> zero-padded LDS tile ----
#pragma unroll
    for (int j = 0; j < pre_a.kPer; ++j) {
      const int i = tid + 256 * j;
      if ((j + 1) * 256 <= kC2 * kM2 || i < kC2 * kM2) {
        const int o = i / kM2, p = i - o * kM2, oy = p / kO2, ox = p - oy * kO2;
        dz2p[o * kZ2 + oy * kZ2W + ox] = pre_a.v[j] > 0.f ? pre_d.v[j] : 0.f;
      }
    }
    // (HAVE_A1) the saved conv1 tile: 39 chunks of 1 KiB by LDS-DMA (ten per wave), no staging registers.  Issued BEHIND
    // the other tiles' LDS stores: the backend cannot tell those stores from the DMA's destination (one dynamic LDS
    // array) and waits for vmcnt(0) in front of the first LDS access after a DMA.  The tile's last 16 floats (channel
    // 15, border row 24) are never written by anybody and stay zero from the initialisation.
    if constexpr (HAVE_A1) {
      const float* srow = a1 + (size_t)n * kA1Row + 4 * lane;
#pragma unroll
      for (int j = 0; j < 10; ++j) {
        const int c = wave + 4 * j;   // wave-uniform
        if (c < kLdsC1 / 256) glds16(srow + 256 * c, c1_pad + 256 * c);
      }
    }
    if constexpr (!HAVE_A1) {   // (with the LDS-DMA in flight the barrier's fence waits for vmcnt(0): the prefetch goes behind it)
      if (n + (int)gridDim.x < n_obs) {
        const size_t nn = (size_t)n + gridDim.x;
        if (words) pre_x.load(reinterpret_cast<const uint32_t*>(obs + nn * 4 * kD * kD), tid);
        pre_a.load(a2 + nn * kC2 * kM2, tid);
        pre_d.load(dy + nn * kC2 * kM2, tid);
      }
    }
    __syncthreads();
    if constexpr (HAVE_A1) {   // the NEXT observation's inputs: in flight during (2) - (4)
      if (n + (int)gridDim.x < n_obs) {
        const size_t nn = (size_t)n + gridDim.x;
        if (words) pre_x.load(reinterpret_cast<const uint32_t*>(obs + nn * 4 * kD * kD), tid);
        pre_a.load(a2 + nn * kC2 * kM2, tid);
        pre_d.load(dy + nn * kC2 * kM2, tid);
      }
    }
    CONV_REGION(1);   // fill
    // ---- (1) conv1 forward into c1_pad (operand values and order of conv12_u8_mfma_kernel) ----
    // Round 4: the phases below spent 3/4 of their time on index arithmetic, not on MFMAs (per-phase clocks of
    // tools/conv_regions.py: 94 k clocks per observation for 23 k clocks of matrix work per wave) — divisions by 21 /
    // 11 / a run-time nx per tile and per output row, a wrap test per position.  Positions now advance
    // incrementally ((1), (3): a tile is 64 positions further = 3 rows + 1 column of 21), and the two weight-gradient
    // loops walk ROWS padded to a multiple of 4 positions ((2): 11 -> 12, (4): 21 -> 24): the padding positions read
    // the zero borders of the dz tiles (A = 0) and any finite B, so they add exact zeros, and every LDS offset
    // inside a row is an immediate.
    if constexpr (!HAVE_A1) {
      int oy = oy1, ox = ox1, ey = ey1, ex = ex1;             // tile t = wave: gather position / first output row of this lane
#pragma unroll 1
      for (int t = wave; t < 28; t += 4) {
        const bool in = oy < kO1;                             // m < 441 (the last tile: clamp to the last position)
        const uint8_t* a_base = in_u8 + (2 * (in ? oy : kO1 - 1)) * kP1 + 2 * (in ? ox : kO1 - 1) + q;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        // all 16 bytes, then all 16 table entries, then the MFMAs (LDS_FENCE: with the register file nearly full
        // the scheduler otherwise issues each read right in front of the MFMA that waits for it)
        uint32_t ub[16];
        float av[16];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) ub[ks] = a_base[(ks >> 2) * kP1 * kP1 + (ks & 3) * kP1];
        LDS_FENCE();
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) av[ks] = lut[ub[ks]];
        LDS_FENCE();
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], bw1[ks], acc, 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);   // the order of the block for the scheduler: byte reads,
        __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);   // table reads,
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);   // MFMAs
        float* crow = c1_pad + col * kP2 * kP2 + (ey + 2) * kP2 + (ex + 2);
#pragma unroll
        for (int r = 0; r < 4; ++r) {                         // rows 4q + r of D: positions (ey, ex + r), wrapping into the next row
          const bool w = ex + r >= kO1;
          const int yy = ey + (w ? 1 : 0);
          if (yy < kO1) {
            const float v = acc[r] + bias1;
            crow[r + (w ? kP2 - kO1 : 0)] = v > 0.f ? v : 0.f;
          }
        }
        oy += 3; ox += 1;                                     // 64 positions further
        if (ox >= kO1) { ox -= kO1; oy += 1; }
        ey += 3; ex += 1;
        if (ex >= kO1) { ex -= kO1; ey += 1; }
      }
    }
    if constexpr (!HAVE_A1) __syncthreads();
    CONV_REGION(2);   // (1) conv1 recompute
    // ---- (2) dW2: this wave owns input channels 4*wave .. 4*wave+3 (k tiles), both o tiles ----
    // 11 rows of 12 positions (ox = 11: dz2p column 11 is zero), three k-steps per row with immediate offsets.
    // The loop over rows stays ROLLED (unrolled the scheduler hoists the LDS gathers of all iterations: > 256 VGPRs).
    {
      const float* za = dz2p + col * kZ2 + q;                 // dz2p[col][oy][4 xs + q]
      const float* bb = c1_pad + (4 * wave) * kP2 * kP2 + kh * kP2 + 2 * q + kw;   // c1_pad[4w + j][2 oy + kh][2 (4 xs + q) + kw]
#pragma clang loop unroll(disable)
      for (int oy = 0; oy < kO2; ++oy) {
#pragma unroll
        for (int xs = 0; xs < 3; ++xs) {
          const float a0 = za[4 * xs], a1v = za[16 * kZ2 + 4 * xs];
          db2a0 += a0;
          db2a1 += a1v;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float b = bb[j * kP2 * kP2 + 8 * xs];
            acc2[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, acc2[0][j], 0, 0, 0);
            acc2[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v, b, acc2[1][j], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 18, 0);   // the row's 18 operand reads, then its 24 MFMAs
        __builtin_amdgcn_sched_group_barrier(0x008, 24, 0);
        za += kZ2W;
        bb += 2 * kP2;
      }
    }
    __syncthreads();   // all patch2 gathers done before dz1 overwrites a1
    CONV_REGION(3);   // (2) dW2
    // ---- (3) dz1 for this wave's parity class, in place over a1 ----
    {
      const int ny = kO2 - py, nx = kO2 - px, M = ny * nx;   // y = 2*iy + py < 21, x = 2*ix + px < 21
      int iy = iy3, ix = ix3, jy = jy3, jx = jx3;             // tile 0: gather position (m = col) / first output row (4 q)
#pragma unroll 1
      for (int t = 0; t * 16 < M; ++t) {
        const bool in = iy < ny;                              // m < M (clamp to the last position)
        const float* ab = dz2p + ((in ? iy : ny - 1) + 1 - ta) * kZ2W + ((in ? ix : nx - 1) + 1 - tb);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, accx = {0.f, 0.f, 0.f, 0.f};
        // the a1 values under this tile's outputs (for the ReLU mask) are read first, the A operands eight k-steps
        // ahead of their MFMAs (LDS_FENCE, see (1))
        float* pz[4];
        float a1v[4];
        bool ok[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {                         // positions (jy, jx + r), wrapping into the next row
          const bool w = jx + r >= nx;
          const int yy = jy + (w ? 1 : 0), xx = jx + r - (w ? nx : 0);
          ok[r] = yy < ny;
          pz[r] = c1_pad + col * kP2 * kP2 + (ok[r] ? (2 * yy + py + 2) * kP2 + (2 * xx + px + 2) : 0);   // [col][0][0]: border
          a1v[r] = *pz[r];
        }
        float av[2][8];
#pragma unroll
        for (int o = 0; o < 8; ++o) av[0][o] = ab[o * kZ2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (g < 3) {
#pragma unroll
            for (int o = 0; o < 8; ++o) av[(g + 1) & 1][o] = ab[(8 * (g + 1) + o) * kZ2];
          }
          LDS_FENCE();
#pragma unroll
          for (int o = 0; o < 8; o += 2) {   // HAVE_A1: two interleaved chains (even / odd o), added below — a single chain waits 40 clocks per MFMA
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g & 1][o], bt[8 * g + o], acc, 0, 0, 0);
            if constexpr (HAVE_A1) accx = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g & 1][o + 1], bt[8 * g + o + 1], accx, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g & 1][o + 1], bt[8 * g + o + 1], acc, 0, 0, 0);
          }
        }
        // the order of the block for the scheduler: the a1 reads + the first eight A operands (ds_read2: two values
        // per instruction), then three times [the next eight operands | eight MFMAs], then the last eight MFMAs
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        if constexpr (HAVE_A1) acc += accx;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (ok[r]) {
            const float v = a1v[r] > 0.f ? acc[r] : 0.f;
            *pz[r] = v;
            db1a += v;
          }
        }
        iy += 1; ix += 16 - nx;                               // 16 positions further (nx = 10 or 11)
        if (ix >= nx) { ix -= nx; iy += 1; }
        jy += 1; jx += 16 - nx;
        if (jx >= nx) { jx -= nx; jy += 1; }
      }
    }
    __syncthreads();
    CONV_REGION(4);   // (3) dz1 (incl. its barrier)
    // ---- (4) dW1: this wave owns input channel `wave` (16 taps = one k tile) ----
    // 21 rows of 24 positions (x = 21 .. 23: the zero border of c1_pad / column 0 of its next row), six k-steps per
    // row with immediate offsets; two accumulators (even / odd k-steps of a row: fixed order, summed at the end).
    {
      const float* pa = c1_pad + col * kP2 * kP2 + 2 * kP2 + 2 + q;     // c1_pad[col][y + 2][4 xs + q + 2]
      const uint8_t* pb = in_u8 + wave * kP1 * kP1 + kh * kP1 + 2 * q + kw;   // in_u8[wave][2 y + kh][2 (4 xs + q) + kw]
      // Two rows per iteration, ping-pong: the operands of the next row are read before the MFMAs of the current one
      // are issued (sched_group_barrier), and the byte becomes (float)u / 255.0f in registers (byte_over_255: the table's
      // values bit for bit) while they run.  22 rows: row 21 reads the zero border rows of c1_pad (A = 0).
      float ca[6], na[6];
      uint32_t cb[6], nb[6];
#pragma unroll
      for (int xs = 0; xs < 6; ++xs) { ca[xs] = pa[4 * xs]; cb[xs] = pb[8 * xs]; }
#pragma clang loop unroll(disable)
      for (int y = 0; y < kO1 + 1; y += 2) {
#pragma unroll
        for (int xs = 0; xs < 6; ++xs) { na[xs] = pa[kP2 + 4 * xs]; nb[xs] = pb[2 * kP1 + 8 * xs]; }
#pragma unroll
        for (int xs = 0; xs < 6; xs += 2) {   // two interleaved chains (even / odd k-steps of a row), added at the end
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[xs], byte_over_255(cb[xs]), acc1, 0, 0, 0);
          if constexpr (HAVE_A1) acc1b = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[xs + 1], byte_over_255(cb[xs + 1]), acc1b, 0, 0, 0);
          else acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[xs + 1], byte_over_255(cb[xs + 1]), acc1, 0, 0, 0);
        }
        pa += 2 * kP2;
        pb += 4 * kP1;
#pragma unroll
        for (int xs = 0; xs < 6; ++xs) { ca[xs] = pa[4 * xs]; cb[xs] = pb[8 * xs]; }
#pragma unroll
        for (int xs = 0; xs < 6; xs += 2) {
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(na[xs], byte_over_255(nb[xs]), acc1, 0, 0, 0);
          if constexpr (HAVE_A1) acc1b = __builtin_amdgcn_mfma_f32_16x16x4f32(na[xs + 1], byte_over_255(nb[xs + 1]), acc1b, 0, 0, 0);
          else acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(na[xs + 1], byte_over_255(nb[xs + 1]), acc1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);    // the order of the block for the scheduler: reads of row
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);    // y + 1, MFMAs of row y, reads of row y + 2, MFMAs of y + 1
        __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
      }
    }
    CONV_REGION(5);   // (4) dW1
  }
  // ---- this workgroup's partial sums ----
#ifdef PARLHIP_CONV_REGIONS
  if (threadIdx.x == 0) atomicAdd(&g_conv_regions[15], (unsigned long long)((n_obs - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x));
#endif
  float* P = partial + (size_t)blockIdx.x * kBwdPartial;
  if constexpr (HAVE_A1) acc1 += acc1b;   // even + odd k-steps (the two interleaved chains of (4))
#pragma unroll
  for (int r = 0; r < 4; ++r) P[kBwdDW1 + (4 * q + r) * kK1 + 16 * wave + col] = acc1[r];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        P[kBwdDW2 + (16 * t + 4 * q + r) * kK2 + 16 * (4 * wave + j) + col] = acc2[t][j][r];
  // bias gradients: per-lane partial row sums -> LDS -> fixed-order sums (deterministic)
  __syncthreads();
  red[tid] = db1a;                         // [wave][q][col]: db1[c = col] = sum over (wave, q)
  if (wave == 0) {                         // every wave saw the same dz2 rows; take wave 0's sums
    red[256 + q * 16 + col] = db2a0;       // [q][col]: db2[o = col]      = sum over q
    red[256 + 64 + q * 16 + col] = db2a1;  //           db2[o = 16 + col]
  }
  __syncthreads();
  if (tid < 16) {
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += red[i * 16 + tid];
    P[kBwdDB1 + tid] = s;
  } else if (tid < 48) {
    const int o = tid - 16;                // 0..31
    const float* r = red + 256 + (o >> 4) * 64 + (o & 15);
    P[kBwdDB2 + o] = r[0] + r[16] + r[32] + r[48];
  }
}

// fixed-order sum of the per-workgroup partials -> dW1 [16,64], db1 [16], dW2 [32,256], db2 [32]
__global__ __launch_bounds__(256) void conv12_bwd_reduce_kernel(const float* __restrict__ partial, int n_parts,
                                                                float* __restrict__ dw1, float* __restrict__ db1,
                                                                float* __restrict__ dw2, float* __restrict__ db2) {
  __shared__ float red[256];
  const int j = blockIdx.x * 16 + (threadIdx.x & 15);
  const float s = partial_sum16(partial, n_parts, kBwdPartial, j, red);
  if (threadIdx.x >= 16 || j >= kBwdPartial) return;
  if (j < kBwdDB1) dw1[j] = s;
  else if (j < kBwdDW2) db1[j - kBwdDB1] = s;
  else if (j < kBwdDB2) dw2[j - kBwdDW2] = s;
  else db2[j - kBwdDB2] = s;
}


// ----------------------------------------------------------------------------------------
// conv1 of the A2C Atari network (the 84x84 -> 20x20 contraction) on the f32 matrix cores.
//
// Reference: examples/A2C/atari_model.py:21-104 (AtariModel): obs / 255, conv1 4->32 k8 s4 p1
// + ReLU (84x84 -> 20x20).  Per observation an implicit GEMM [400 positions x 256] x [256 x 32]
// (6.55 MFLOP), k = c*64 + kh*8 + kw (the order of weight.flatten(1)).
//
// One workgroup (4 wavefronts) per observation, grid-stride over observations:
//   * the observation stays uint8 in LDS, UNSHIFTED (tile[c][y][x] at byte 88 + c*7056 + y*84 + x, copied with
//     4-byte loads / stores): 28.3 KB instead of the 112.9 KB of a float tile, so that the kernel co-resides
//     with the learner's kernels (75-113 KB) and with two more workgroups of its own (round 3; the float
//     tile was exclusive on its CU).  An A element is u / 255 computed in registers as q = u * r,
//     q += fma(-q, 255, u) * r with r = 1 / 255.0f — bit-identical to the IEEE division for all 256 bytes
//     (checked exhaustively on the host and by the kernel's tests), four VALU operations in the shadow of the
//     two MFMAs they feed, and no second dependent LDS read (the table form the backward kernels use).
//     The padding costs no branch: pad 1 with floor((84+2-8)/4)+1 = 20 outputs reads input rows / columns
//     -1 .. 82, never 83.  Row 83 and column 83 of every plane are stored as zeros, and in the flat layout
//     "column -1 of row y" IS column 83 of row y-1, "row -1 of plane c" IS row 83 of plane c-1; plane 0 has
//     88 zero bytes in front of it for the same purpose.
//   * the whole B operand (the 32 KB weight matrix) lives in registers: 64 k-steps x 2 N-tiles =
//     128 VGPRs per lane, loaded once per workgroup;
//   * each wave owns M-tiles (16 output positions) and issues 2 MFMAs per A gather (both N-tiles);
//   * D (col = channel, rows = 4 consecutive positions) + bias + ReLU goes straight to HBM as one
//     16-byte store per lane in NCHW order — no staging, no im2col matrix in HBM (the GEMM-lowered
//     conv writes and re-reads 410 KB of patches per observation).
// Algorithmic bytes per observation: 28,224 read + 51,200 written.
// ----------------------------------------------------------------------------------------
constexpr int kD84 = 84;                 // input size
constexpr int kO84 = 20, kC84 = 32;      // conv1 output size / channels
constexpr int kM84 = kO84 * kO84;        // 400 positions = 25 M-tiles of 16
constexpr int kK84 = 4 * 8 * 8;          // 256
constexpr int kPlane84 = kD84 * kD84;    // 7056
constexpr int kGuard84 = 88;                                     // zero bytes in front of plane 0 (>= 85, 4-byte multiple)
constexpr int kLds84u8Bytes = kGuard84 + 4 * kPlane84 + 8;       // 28,320

// RING: the four 84x84 frames of env n read from the rollout ring in place (see RingObs).  PACKED: `w` is the weight
// matrix in operand order, wt1[ks][nt][lane] (the layout conv23_84_mfma_kernel's wt2 / wt3 have): 128 coalesced
// 256-byte loads per wave instead of 128 loads touching 16 cache lines each — the start-up cost conv12_u8_mfma_kernel
// had (see there).
template <bool RING, bool PACKED>
__global__ __launch_bounds__(256, 2) void conv1_84_u8_mfma_kernel(
    const uint8_t* __restrict__ obs, RingObs ro, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ out, int n_obs) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds8[];
  uint8_t* tile = lds8;                                          // [88 guard][4][84][84]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, col = lane & 15;
  // B[k][n] = w[n][k]: lane (q, col) holds k = 4*ks + q, n = 16*nt + col
  float breg[64][2];
#pragma unroll
  for (int ks = 0; ks < 64; ++ks) {
    if constexpr (PACKED) {
      breg[ks][0] = w[(ks * 2) * 64 + lane];
      breg[ks][1] = w[(ks * 2 + 1) * 64 + lane];
    } else {
      breg[ks][0] = w[col * kK84 + ks * 4 + q];
      breg[ks][1] = w[(16 + col) * kK84 + ks * 4 + q];
    }
  }
  const float bias0 = bias[col], bias1 = bias[16 + col];
  if (tid < kGuard84 / 4) reinterpret_cast<uint32_t*>(tile)[tid] = 0u;
  Batch<kPlane84, uint32_t> pre;   // 4 * 7056 bytes = 7056 words, 28 per thread: the NEXT observation (see Batch)
  auto fetch = [&](int n) {
    if constexpr (RING) {
      const int sr = ro.since[(size_t)ro.slot * ro.E + n];           // wave-uniform
      const uint32_t* fr[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int from = ro.slot - (3 - j < sr ? 3 - j : sr);
        from += from < 0 ? ro.num_slots : 0;
        fr[j] = reinterpret_cast<const uint32_t*>(ro.ring + ((size_t)from * ro.E + n) * kPlane84);   // 7056 B: a word multiple
      }
      pre.load_by(tid, [&](int wi) {
        const int c = wi / (kPlane84 / 4);
        const uint32_t* f = c == 0 ? fr[0] : (c == 1 ? fr[1] : (c == 2 ? fr[2] : fr[3]));
        return f + (wi - c * (kPlane84 / 4));
      });
    } else {
      pre.load(reinterpret_cast<const uint32_t*>(obs + (size_t)n * 4 * kPlane84), tid);
    }
  };
  if ((int)blockIdx.x < n_obs) fetch(blockIdx.x);
  for (int n = blockIdx.x; n < n_obs; n += gridDim.x) {
    __syncthreads();  // the previous observation's gathers are done before the tile is rewritten
    uint32_t* dstw = reinterpret_cast<uint32_t*>(tile + kGuard84);
    pre.each(tid, [&](int wi, uint32_t v) {  // 84 % 4 == 0: a word never spans two rows
      const int i = wi * 4;
      const int r = i % kPlane84, y = r / kD84, x = r - y * kD84;
      v = (x == kD84 - 4) ? (v & 0x00ffffffu) : v;   // column 83 := 0 (it doubles as column -1 of the next row)
      v = (y == kD84 - 1) ? 0u : v;                  // row 83 := 0 (it doubles as row -1 of the next plane)
      dstw[wi] = v;
    });
    if (n + (int)gridDim.x < n_obs) fetch(n + gridDim.x);
    __syncthreads();
    float* dst = out + (size_t)n * kC84 * kM84;
    for (int mt = wave; mt < kM84 / 16; mt += 4) {
      const int m = mt * 16 + col;
      const int oy = m / kO84, ox = m - oy * kO84;
      // input (4 oy + kh - 1, 4 ox + kw - 1), kw = (ks & 1) * 4 + q
      const uint8_t* a_base = tile + kGuard84 + (4 * oy - 1) * kD84 + (4 * ox - 1) + q;
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      // k = 4*ks + q = c*64 + kh*8 + kw  ->  c = ks>>4, kh = (ks>>1)&7, kw = (ks&1)*4 + q.  The bytes of the NEXT
      // 16 k-steps (one input plane) are in flight while the MFMAs of this plane run: without the explicit
      // double buffer the compiler issues every LDS read right in front of its use.
      uint32_t ub[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) ub[j] = a_base[((j >> 1) & 7) * kD84 + (j & 1) * 4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float a[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) a[j] = byte_over_255(ub[j]);
        if (c < 3) {
#pragma unroll
          for (int j = 0; j < 16; ++j) ub[j] = a_base[(c + 1) * kPlane84 + ((j >> 1) & 7) * kD84 + (j & 1) * 4];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], breg[c * 16 + j][0], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], breg[c * 16 + j][1], acc1, 0, 0, 0);
        }
      }
      // D: column = channel (col), rows 4q..4q+3 = 4 consecutive positions -> one 16 B store
      f32x4 o0, o1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v0 = acc0[r] + bias0, v1 = acc1[r] + bias1;
        o0[r] = v0 > 0.f ? v0 : 0.f;
        o1[r] = v1 > 0.f ? v1 : 0.f;
      }
      const int mo = mt * 16 + q * 4;
      *reinterpret_cast<f32x4*>(dst + col * kM84 + mo) = o0;
      *reinterpret_cast<f32x4*>(dst + (16 + col) * kM84 + mo) = o1;
    }
  }
}


// ----------------------------------------------------------------------------------------
// conv2 + conv3 of the A2C Atari network, fused (examples/A2C/atari_model.py:21-104):
//   a1 [32,20,20] (the output of conv1_84_u8_mfma_kernel) -> conv2 32->64 k4 s2 p2 + ReLU
//   -> a2 [64,11,11] (stays in LDS) -> conv3 64->64 k3 s1 + ReLU -> a3 [64,9,9] = the 5184-wide
//   input of the fc layer.  13.9 MFLOP per observation.
// One workgroup per observation (grid-stride).  conv2 sums over its 32 input channels in channel-major k order,
// so a1 goes through LDS EIGHT CHANNELS AT A TIME (round 4): a zero-padded [8][24][24] f32 tile (18.4 KB), four
// passes of 32 k-steps over the same accumulators — the same sequence of FMAs as with the whole [32][24][24] tile
// (73.7 KB, rounds 1-3): bit-identical outputs — the next quarter's 12.5 floats per thread prefetched into
// registers while the MFMAs of the current one run.  With a2 in LDS ([64][121], 31 KB) that is 49.4 KB instead of
// 104.7: the ACTORS' conv2 + conv3 now fits on a CU beside a learner workgroup (75-113 KB) instead of waiting for the
// learner's persistent grids to end — at 84x84 rollout and update used to exclude each other almost completely
// (rollout alone 55 ms + update alone 56 ms = 112-115 ms overlapped).
// The weight matrices (128 KB + 147 KB) fit neither registers nor
// the remaining LDS: they are STREAMED from L2 in MFMA operand order — `wt2[ks][nt][lane]`,
// `wt3[ks][nt][lane]` (prepared by the host wrapper: one 256-byte coalesced load per wave and
// k-step) — each wave owning one 16-channel N tile and all M tiles, so every streamed B value
// feeds 8 (conv2) / 6 (conv3) MFMAs.  k order: conv2 k = c*16 + kh*4 + kw (natural, kw = lane
// quarter); conv3 k' = (kh*3 + kw)*64 + c (tap-major, so the tap is uniform per k-step).
// a2 is also written to HBM when the learner needs it for the backward pass.
// ----------------------------------------------------------------------------------------
constexpr int kA1 = 20, kA1P = 24, kA1Plane = kA1P * kA1P;      // conv1 output, padded by 2
constexpr int kA2 = 11, kM2b = kA2 * kA2;                        // 121 conv2 outputs
constexpr int kA3 = 9, kM3 = kA3 * kA3;                          // 81 conv3 outputs
constexpr int kA1Q = 8;                                           // a1 channels per LDS pass
// The streamed B operand (one coalesced 256-byte load per wave and k-step, from L2) is loaded kBPf k-steps before
// the MFMAs that consume it: with the load and its `s_waitcnt vmcnt` in the same iteration (rounds 1-3: two k-steps
// per iteration) every 16 MFMAs (512 clocks of the matrix pipe) waited out one L2 round trip of about the same
// length — the ~40 % of the f32 MFMA peak these kernels measured, one wave per SIMD and nobody to hide it.  Same
// FMAs in the same order: bit-identical.  (16 % kBPf == 0: a conv3 tap never changes inside a group.)
#ifndef PARLHIP_BPF
#define PARLHIP_BPF 8
#endif
constexpr int kBPf = PARLHIP_BPF;      // the backward kernels (one wave per SIMD by their accumulators anyway)
constexpr int kBPfFwd = 4;             // conv23_84_mfma_kernel: 128 VGPRs, two of its waves fit beside an emulator wave pair
constexpr int kLds23Floats = kA1Q * kA1Plane + 64 * kM2b;        // 4,608 + 7,744 = 12,352 floats = 49,408 B

__global__ __launch_bounds__(256) void conv23_84_mfma_kernel(
    const float* __restrict__ a1, const float* __restrict__ wt2, const float* __restrict__ b2,
    const float* __restrict__ wt3, const float* __restrict__ b3, float* __restrict__ a2_out,
    float* __restrict__ a3_out, int n_obs) {
  extern __shared__ float lds[];
  float* a1p = lds;                       // [8][24][24]: one quarter of a1's channels at a time
  float* a2s = lds + kA1Q * kA1Plane;     // [64][121]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, col = lane & 15;
  const float bias2 = b2[16 * wave + col], bias3 = b3[16 * wave + col];
  for (int i = tid; i < kA1Q * kA1Plane; i += 256) a1p[i] = 0.0f;   // borders stay zero
  // conv2 gather offsets of this lane's 8 M tiles: position m -> (2*oy)*24 + 2*ox + kw (kw = q)
  int off2[8];
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) {
    int m = mt * 16 + col;
    m = m < kM2b ? m : kM2b - 1;
    const int oy = m / kA2, ox = m - oy * kA2;
    off2[mt] = (2 * oy) * kA1P + 2 * ox + q;
  }
  int off3[6];
#pragma unroll
  for (int mt = 0; mt < 6; ++mt) {
    int m = mt * 16 + col;
    m = m < kM3 ? m : kM3 - 1;
    const int oy = m / kA3, ox = m - oy * kA3;
    off3[mt] = oy * kA2 + ox + q * kM2b;   // + channel (c = 4*(ks&15) + q) and tap offsets per k-step
  }
#pragma unroll 1
  for (int n = blockIdx.x; n < n_obs; n += gridDim.x) {
    // ---- conv2: wave = N tile (channels 16*wave ..), 8 M tiles, 128 k-steps = 4 passes of 8 input channels ----
    {
      constexpr int kQ4 = kA1Q * kA1 * kA1 / 4;   // 800 float4 per quarter: up to 4 per thread
      const float4* src = reinterpret_cast<const float4*>(a1 + (size_t)n * 32 * kA1 * kA1);
      float4 pre[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { const int i = tid + 256 * j; if (i < kQ4) pre[j] = src[i]; }
      f32x4 acc[8];
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* wp = wt2 + wave * 64 + lane;            // wt2[ks][nt = wave][lane]
      float nb[kBPfFwd];                                      // the B operand, kBPfFwd k-steps ahead (see kBPfFwd)
#pragma unroll
      for (int j = 0; j < kBPfFwd; ++j) nb[j] = wp[j * 256];
#pragma unroll 1
      for (int pass = 0; pass < 32 / kA1Q; ++pass) {
        __syncthreads();   // the previous pass (or the previous observation's conv3) is done with the tile
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = tid + 256 * j;
          if (i < kQ4) {
            const int e = i * 4, c = e / (kA1 * kA1), r = e - c * kA1 * kA1, y = r / kA1, x = r - y * kA1;   // 20 % 4 == 0
            float* d = a1p + c * kA1Plane + (y + 2) * kA1P + (x + 2);
            d[0] = pre[j].x; d[1] = pre[j].y; d[2] = pre[j].z; d[3] = pre[j].w;
          }
        }
        __syncthreads();
        if (pass + 1 < 32 / kA1Q) {   // the next quarter is in flight while this one's MFMAs run
#pragma unroll
          for (int j = 0; j < 4; ++j) { const int i = tid + 256 * j; if (i < kQ4) pre[j] = src[(pass + 1) * kQ4 + i]; }
        }
#pragma unroll 1
        for (int kq0 = 0; kq0 < 4 * kA1Q; kq0 += kBPfFwd) {
          float cb[kBPfFwd];
#pragma unroll
          for (int j = 0; j < kBPfFwd; ++j) cb[j] = nb[j];
          const int kn = pass * 4 * kA1Q + kq0 + kBPfFwd;
          if (kn < 128) {
#pragma unroll
            for (int j = 0; j < kBPfFwd; ++j) nb[j] = wp[(kn + j) * 256];
          }
#pragma unroll
          for (int j = 0; j < kBPfFwd; ++j) {
            const int kq = kq0 + j;
            const float* ab = a1p + (kq >> 2) * kA1Plane + (kq & 3) * kA1P;   // c = 8 * pass + (kq >> 2), kh = kq & 3
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ab[off2[mt]], cb[j], acc[mt], 0, 0, 0);
          }
        }
      }
      float* g2 = a2_out ? a2_out + (size_t)n * 64 * kM2b : nullptr;
#pragma unroll
      for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int mo = mt * 16 + q * 4 + r;
          if (mo < kM2b) {
            float v = acc[mt][r] + bias2;
            v = v > 0.f ? v : 0.f;
            a2s[(16 * wave + col) * kM2b + mo] = v;
            if (g2) g2[(16 * wave + col) * kM2b + mo] = v;
          }
        }
    }
    __syncthreads();
    // ---- conv3: wave = N tile, 6 M tiles, 144 k-steps in tap-major order ----
    {
      f32x4 acc[6];
#pragma unroll
      for (int mt = 0; mt < 6; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* wp = wt3 + wave * 64 + lane;
      float nb[kBPfFwd];
#pragma unroll
      for (int j = 0; j < kBPfFwd; ++j) nb[j] = wp[j * 256];
#pragma unroll 1
      for (int ks0 = 0; ks0 < 144; ks0 += kBPfFwd) {
        float cb[kBPfFwd];
#pragma unroll
        for (int j = 0; j < kBPfFwd; ++j) cb[j] = nb[j];
        if (ks0 + kBPfFwd < 144) {
#pragma unroll
          for (int j = 0; j < kBPfFwd; ++j) nb[j] = wp[(ks0 + kBPfFwd + j) * 256];
        }
        const int tap = ks0 >> 4, kh = tap / 3, kw = tap - kh * 3;   // k' = tap*64 + c, c = 4*(ks & 15) + q  (16 % kBPfFwd == 0)
#pragma unroll
        for (int j = 0; j < kBPfFwd; ++j) {
          const float* ab = a2s + (4 * ((ks0 + j) & 15)) * kM2b + kh * kA2 + kw;
#pragma unroll
          for (int mt = 0; mt < 6; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ab[off3[mt]], cb[j], acc[mt], 0, 0, 0);
        }
      }
      float* g3 = a3_out + (size_t)n * 64 * kM3;
#pragma unroll
      for (int mt = 0; mt < 6; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int mo = mt * 16 + q * 4 + r;
          if (mo < kM3) {
            const float v = acc[mt][r] + bias3;
            g3[(16 * wave + col) * kM3 + mo] = v > 0.f ? v : 0.f;
          }
        }
    }
  }
}


// ----------------------------------------------------------------------------------------
// Backward of the 84x84 model's convolutions for the learner (A2C.learn / IMPALA.learn on
// examples/A2C/atari_model.py:21-104), layer by layer, one workgroup per observation, the
// activations a1 / a2 / a3 saved by the forward kernels (51 + 31 + 21 KB per observation: HBM traffic
// of the whole backward pass ~0.3 MB per observation against 41 MFLOP of MFMA work).  Weight
// gradients accumulate in registers across the observations of a workgroup and leave as
// per-workgroup partials summed in a fixed order (partial_sum_kernel): deterministic.
//
// conv3_84_bwd_kernel: dz3 = dy3 * (a3 > 0);
//   (A) dW3[o][k'] += sum_p dz3[o][p] a2[c][oy+kh][ox+kw]       [64 x 81] x [81 x 576], k' = tap*64 + c
//       wave w owns the k' tiles {w, w+4, ..} (9 tiles x 4 o tiles = 36 accumulators)
//   (B) dA2[c][y][x] = sum_{o,kh,kw} dz3[o][y-kh][x-kw] w3[o][c][kh][kw]   (gather from dz3 padded
//       by 2) = [121 x 576] x [576 x 64] with the B operand streamed (wt3b[ks][nt][lane],
//       k'' = tap*64 + o); dz2 = dA2 * (a2 > 0) -> HBM
// ----------------------------------------------------------------------------------------
constexpr int kZ3P = 13, kZ3Plane = kZ3P * kZ3P;                      // dz3 padded by 2: 13 x 13
constexpr int kLds3bFloats = 64 * kM2b + 64 * kZ3Plane + 128;          // 7,744 + 10,816 + 128 = 18,688 floats = 74,752 B
constexpr int kPart3 = 64 * 576 + 64;                                  // dW3 [o][k'] + db3

// PARLHIP_C84_COEX (experiment): <= 256 registers per lane so that two env waves (2 x 128) fit beside this wave on a SIMD;
// >= 2: the 84x84 kernels that run two workgroups per CU run one (kC84PerCU)
#if defined(PARLHIP_C84_COEX) && PARLHIP_C84_COEX >= 2
constexpr int kC84PerCU = 1;
#else
constexpr int kC84PerCU = 2;
#endif
// PARLHIP_C84_LEARNER_PER_CU (experiment): the big launches (>= 4096 observations: the learner's) of the 84x84 kernels
// that run two workgroups per CU
#ifndef PARLHIP_C84_LEARNER_PER_CU
#define PARLHIP_C84_LEARNER_PER_CU kC84PerCU
#endif
#ifndef PARLHIP_C84_LEARNER_MIN_OBS
#define PARLHIP_C84_LEARNER_MIN_OBS 4096
#endif
static inline int c84_per_cu(int n_obs) { return n_obs >= PARLHIP_C84_LEARNER_MIN_OBS ? PARLHIP_C84_LEARNER_PER_CU : kC84PerCU; }
#ifdef PARLHIP_C84_COEX
#define PARLHIP_C84_BWD_BOUNDS __launch_bounds__(256, 2)
#else
#define PARLHIP_C84_BWD_BOUNDS __launch_bounds__(256)
#endif
__global__ PARLHIP_C84_BWD_BOUNDS void conv3_84_bwd_kernel(
    const float* __restrict__ a2, const float* __restrict__ a3, const float* __restrict__ dy3,
    const float* __restrict__ wt3b, float* __restrict__ dz2, float* __restrict__ partial, int n_obs) {
  extern __shared__ float lds[];
  float* a2s = lds;                         // [64][121]
  float* z3p = lds + 64 * kM2b;             // [64][13][13], zero border
  float* red = z3p + 64 * kZ3Plane;         // [128]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, col = lane & 15;
  f32x4 accw[4][9];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int j = 0; j < 9; ++j) accw[mt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dba[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < 64 * kZ3Plane; i += 256) z3p[i] = 0.0f;
  // (A) B-operand offsets of this lane's 9 k' tiles: tile nt = wave + 4 j -> tap = nt >> 2, c = 16 (nt & 3) + col
  int boff[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    const int nt = wave + 4 * j, tap = nt >> 2, kh = tap / 3, kw = tap - kh * 3;
    boff[j] = (16 * (nt & 3) + col) * kM2b + kh * kA2 + kw;
  }
  // (B) A-operand offsets of the 8 M tiles (positions of the 11 x 11 input): z3p[o][y + 2][x + 2]
  int aoff[8];
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) {
    int m = mt * 16 + col;
    m = m < kM2b ? m : kM2b - 1;
    const int y = m / kA2, x = m - y * kA2;
    aoff[mt] = (y + 2) * kZ3P + (x + 2) + q * kZ3Plane;    // + o = 4 (ks & 15) + q, - kh * 13 - kw per k-step
  }
  Batch<64 * kM2b / 4, float4> pre_a2;      // 7,744 floats per observation (a 16-byte multiple): the NEXT observation (see Batch)
  Batch<64 * kM3, float> pre_a3, pre_dy;
  if ((int)blockIdx.x < n_obs) {
    pre_a2.load(reinterpret_cast<const float4*>(a2 + (size_t)blockIdx.x * 64 * kM2b), tid);
    pre_a3.load(a3 + (size_t)blockIdx.x * 64 * kM3, tid);
    pre_dy.load(dy3 + (size_t)blockIdx.x * 64 * kM3, tid);
  }
#pragma unroll 1
  for (int n = blockIdx.x; n < n_obs; n += gridDim.x) {
    __syncthreads();
    pre_a2.each(tid, [&](int i, float4 v) { reinterpret_cast<float4*>(a2s)[i] = v; });
#pragma unroll
    for (int j = 0; j < pre_a3.kPer; ++j) {
      const int i = tid + 256 * j;
      if ((j + 1) * 256 <= 64 * kM3 || i < 64 * kM3) {
        const int o = i / kM3, p = i - o * kM3, y = p / kA3, x = p - y * kA3;
        z3p[o * kZ3Plane + (y + 2) * kZ3P + (x + 2)] = pre_a3.v[j] > 0.f ? pre_dy.v[j] : 0.f;
      }
    }
    if (n + (int)gridDim.x < n_obs) {
      const size_t nn = (size_t)n + gridDim.x;
      pre_a2.load(reinterpret_cast<const float4*>(a2 + nn * 64 * kM2b), tid);
      pre_a3.load(a3 + nn * 64 * kM3, tid);
      pre_dy.load(dy3 + nn * 64 * kM3, tid);
    }
    __syncthreads();
    // ---- (A) dW3 ----
    {
      int oy = 0, ox = q;                                  // p = 4 ks + q
#pragma clang loop unroll(disable)
      for (int ks = 0; ks < 21; ++ks) {
        const bool valid = (ks < 20) | (q == 0);           // p < 81
        const int zo = valid ? (oy + 2) * kZ3P + (ox + 2) : 0;   // z3p[o][0][0] is border: 0
        const int po = valid ? oy * kA2 + ox : 0;
        float av[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          av[mt] = z3p[(16 * mt + col) * kZ3Plane + zo];
          dba[mt] += av[mt];
        }
#pragma unroll
        for (int j = 0; j < 9; ++j) {
          const float b = a2s[boff[j] + po];
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) accw[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], b, accw[mt][j], 0, 0, 0);
        }
        ox += 4;
        const bool wrap = ox >= kA3;
        ox -= wrap ? kA3 : 0;
        oy += wrap ? 1 : 0;
      }
    }
    // ---- (B) dz2 = (transposed conv3 of dz3) * (a2 > 0) ----
    {
      f32x4 acc[8];
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* wp = wt3b + wave * 64 + lane;
      float nb[kBPf];
#pragma unroll
      for (int j = 0; j < kBPf; ++j) nb[j] = wp[j * 256];
#pragma unroll 1
      for (int ks0 = 0; ks0 < 144; ks0 += kBPf) {
        float cb[kBPf];
#pragma unroll
        for (int j = 0; j < kBPf; ++j) cb[j] = nb[j];
        if (ks0 + kBPf < 144) {
#pragma unroll
          for (int j = 0; j < kBPf; ++j) nb[j] = wp[(ks0 + kBPf + j) * 256];
        }
        const int tap = ks0 >> 4, kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
        for (int j = 0; j < kBPf; ++j) {
          const float* ab = z3p + (4 * ((ks0 + j) & 15)) * kZ3Plane - kh * kZ3P - kw;
#pragma unroll
          for (int mt = 0; mt < 8; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ab[aoff[mt]], cb[j], acc[mt], 0, 0, 0);
        }
      }
      float* g = dz2 + (size_t)n * 64 * kM2b;
#pragma unroll
      for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int mo = mt * 16 + q * 4 + r;
          if (mo < kM2b) {
            const int idx = (16 * wave + col) * kM2b + mo;
            g[idx] = a2s[idx] > 0.f ? acc[mt][r] : 0.f;
          }
        }
    }
  }
  // ---- partials: dW3[o][k'] (o = 16 mt + 4 q + r, k' = 16 (wave + 4 j) + col), db3 ----
  float* P = partial + (size_t)blockIdx.x * kPart3;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) P[(16 * mt + 4 * q + r) * 576 + 16 * (wave + 4 * j) + col] = accw[mt][j][r];
  __syncthreads();
  if (wave == 0) {   // every wave accumulated the same dz3 row sums
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) red[(mt * 4 + q) * 16 + col] = dba[mt];   // [mt][q][col]
  }
  __syncthreads();
  if (tid < 64) {
    const int mt = tid >> 4, c = tid & 15;
    P[64 * 576 + tid] = red[(mt * 4 + 0) * 16 + c] + red[(mt * 4 + 1) * 16 + c] + red[(mt * 4 + 2) * 16 + c] +
                        red[(mt * 4 + 3) * 16 + c];
  }
}

// conv2_84_bwd_kernel: given a1 [32,20,20] and dz2 [64,11,11] (already masked by a2 > 0):
//   (A) dW2[o][k] += sum_p dz2[o][p] a1pad[c][2 oy + kh][2 ox + kw]    [64 x 121] x [121 x 512], k = c*16 + kh*4 + kw
//       wave w owns the input channels {w, w+4, ..} (8 k tiles x 4 o tiles = 32 accumulators)
//   (B) dA1 = the transposed convolution as a gather per output parity class (stride 2, kernel 4, as
//       in conv12_bwd_u8_mfma_kernel): y = 2 iy + py receives from dz2[o][iy+1-a][ix+1-b] through
//       taps (py+2a, px+2b); one class per wave: [100 x 256] x [256 x 32], B operand (class slice of
//       w2, 32 KB) streamed: wt2b[class][ks][nt][lane], k = (o, a, b) = 4 ks + q; dz1 = dA1 * (a1 > 0) -> HBM
constexpr int kLds2bFloats = 32 * kA1Plane + 64 * kM2b + 128;            // 18,432 + 7,744 + 128 = 26,304 floats = 105,216 B
constexpr int kPart2 = 64 * 512 + 64;

__global__ PARLHIP_C84_BWD_BOUNDS void conv2_84_bwd_kernel(
    const float* __restrict__ a1, const float* __restrict__ dz2, const float* __restrict__ wt2b,
    float* __restrict__ dz1, float* __restrict__ partial, int n_obs) {
  extern __shared__ float lds[];
  float* a1p = lds;                       // [32][24][24], zero border
  float* z2s = lds + 32 * kA1Plane;       // [64][11][11]
  float* red = z2s + 64 * kM2b;           // [128]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, col = lane & 15;
  f32x4 accw[4][8];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int j = 0; j < 8; ++j) accw[mt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dba[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < 32 * kA1Plane; i += 256) a1p[i] = 0.0f;
  const int kh = col >> 2, kw = col & 3;
  // (B) this wave's parity class; A-operand offsets of its 7 M tiles (100 positions iy, ix < 10)
  const int py = wave >> 1, px = wave & 1, ta = q >> 1, tb = q & 1;
  int aoff[7];
#pragma unroll
  for (int mt = 0; mt < 7; ++mt) {
    int m = mt * 16 + col;
    m = m < 100 ? m : 99;
    const int iy = m / 10, ix = m - iy * 10;
    aoff[mt] = (iy + 1 - ta) * kA2 + (ix + 1 - tb);        // z2s[o][iy+1-a][ix+1-b], + o * 121 per k-step
  }
#pragma unroll 1
  for (int n = blockIdx.x; n < n_obs; n += gridDim.x) {
    {  // (no room in the register file to hold the NEXT observation across the MFMA phases: batches of this one)
      Batch<32 * kA1 * kA1 / 4, float4> ba;
      Batch<64 * kM2b / 4, float4> bz;
      ba.load(reinterpret_cast<const float4*>(a1 + (size_t)n * 32 * kA1 * kA1), tid);
      bz.load(reinterpret_cast<const float4*>(dz2 + (size_t)n * 64 * kM2b), tid);
      __syncthreads();   // the loads are in flight while the other waves finish the previous observation
      ba.each(tid, [&](int i, float4 v) {
        const int e = i * 4, c = e / (kA1 * kA1), r = e - c * kA1 * kA1, y = r / kA1, x = r - y * kA1;   // 20 % 4 == 0
        float* d = a1p + c * kA1Plane + (y + 2) * kA1P + (x + 2);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      });
      bz.each(tid, [&](int i, float4 v) { reinterpret_cast<float4*>(z2s)[i] = v; });
    }
    __syncthreads();
    // ---- (A) dW2 ----
    {
      int ox = q, zoff = q, boff = kh * kA1P + 2 * q + kw;   // p = 4 ks + q: (oy, ox) = (0, q)
#pragma clang loop unroll(disable)
      for (int ks = 0; ks < 31; ++ks) {
        const bool valid = (ks < 30) | (q == 0);             // p < 121
        float av[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const float v = z2s[(16 * mt + col) * kM2b + (valid ? zoff : 0)];
          av[mt] = valid ? v : 0.f;
          dba[mt] += av[mt];
        }
        const float* bb = a1p + wave * kA1Plane + (valid ? boff : 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float b = bb[(4 * j) * kA1Plane];            // input channel c = wave + 4 j
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) accw[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], b, accw[mt][j], 0, 0, 0);
        }
        ox += 4;
        const bool wrap = ox >= kA2;
        ox -= wrap ? kA2 : 0;
        zoff += 4;                                           // z2s rows are contiguous: p itself
        boff += wrap ? 8 + (2 * kA1P - 2 * kA2) : 8;         // two rows down, 22 columns back
      }
    }
    // ---- (B) dz1 for this wave's parity class ----
    {
      f32x4 acc[7][2];
#pragma unroll
      for (int mt = 0; mt < 7; ++mt) { acc[mt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[mt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      const float* wp = wt2b + (size_t)wave * 64 * 2 * 64 + lane;   // wt2b[class][ks][nt][lane]
      float nb[kBPf];
#pragma unroll
      for (int j = 0; j < kBPf; ++j) nb[j] = wp[j * 64];
#pragma unroll 1
      for (int o0 = 0; o0 < 64; o0 += kBPf / 2) {
        float cb[kBPf];
#pragma unroll
        for (int j = 0; j < kBPf; ++j) cb[j] = nb[j];
        if (o0 + kBPf / 2 < 64) {
#pragma unroll
          for (int j = 0; j < kBPf; ++j) nb[j] = wp[((o0 + kBPf / 2) * 2 + j) * 64];
        }
#pragma unroll
        for (int j = 0; j < kBPf / 2; ++j) {
          const float* ab = z2s + (o0 + j) * kM2b;
#pragma unroll
          for (int mt = 0; mt < 7; ++mt) {
            const float a = ab[aoff[mt]];
            acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, cb[2 * j], acc[mt][0], 0, 0, 0);
            acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, cb[2 * j + 1], acc[mt][1], 0, 0, 0);
          }
        }
      }
      float* g = dz1 + (size_t)n * 32 * kA1 * kA1;
#pragma unroll
      for (int mt = 0; mt < 7; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int mo = mt * 16 + q * 4 + r;
          if (mo < 100) {
            const int jy = mo / 10, jx = mo - jy * 10, y = 2 * jy + py, x = 2 * jx + px;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
              const int c = 16 * nt + col;
              g[c * kA1 * kA1 + y * kA1 + x] = a1p[c * kA1Plane + (y + 2) * kA1P + (x + 2)] > 0.f ? acc[mt][nt][r] : 0.f;
            }
          }
        }
    }
  }
  // ---- partials: dW2[o][k] (o = 16 mt + 4 q + r, k = 16 (wave + 4 j) + col), db2 ----
  float* P = partial + (size_t)blockIdx.x * kPart2;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) P[(16 * mt + 4 * q + r) * 512 + 16 * (wave + 4 * j) + col] = accw[mt][j][r];
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) red[(mt * 4 + q) * 16 + col] = dba[mt];
  }
  __syncthreads();
  if (tid < 64) {
    const int mt = tid >> 4, c = tid & 15;
    P[64 * 512 + tid] = red[(mt * 4 + 0) * 16 + c] + red[(mt * 4 + 1) * 16 + c] + red[(mt * 4 + 2) * 16 + c] +
                        red[(mt * 4 + 3) * 16 + c];
  }
}

// conv1_84_bwd_kernel: dW1[c][k] += sum_p dz1[c][p] x[ci][4 oy + kh - 1][4 ox + kw - 1] / 255
//   [32 x 400] x [400 x 256], k = ci*64 + kh*8 + kw; the observation stays uint8 in LDS (a tile
//   shifted by the padding: tile[c][py][px] = obs[c][py-1][px-1], row 0 / column 0 = 0; input row / column 83
//   lie outside every window); wave w owns the k tiles {w, w+4, w+8, w+12} x 2 c tiles.  (The (float)u / 255.0f table
//   of rounds 1-3 still has its 1 KB in the LDS layout; since round 4 the byte is converted in registers, see the loop.)
constexpr int kLds1bFloats = (4 * kPlane84) / 4 + 256 + 32 * kM84 + 64;   // 7,056 + 256 + 12,800 + 64 floats = 80,704 B
constexpr int kPart1 = 32 * 256 + 32;

__global__ __launch_bounds__(256, 2) void conv1_84_bwd_kernel(
    const uint8_t* __restrict__ obs, const float* __restrict__ dz1, float* __restrict__ partial, int n_obs) {
  extern __shared__ float lds[];
  uint8_t* tile = reinterpret_cast<uint8_t*>(lds);        // [4][84][84] uint8, shifted by the padding
  float* lut = lds + kPlane84;                            // [256]
  float* z1s = lut + 256;                                 // [32][400]
  float* red = z1s + 32 * kM84;                           // [64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, col = lane & 15;
  f32x4 accw[2][4];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int j = 0; j < 4; ++j) accw[mt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dba[2] = {0.f, 0.f};
  for (int i = tid; i < kPlane84; i += 256) lds[i] = 0.0f;   // the tile incl. its zero row 0 / column 0
  __syncthreads();
  lut[tid] = (float)tid / 255.0f;
  // B-operand offsets of this lane's 4 k tiles: nt = wave + 4 j -> ci = nt >> 2, kh = 2 (nt & 3) + (col >> 3), kw = col & 7
  int boff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int nt = wave + 4 * j;
    boff[j] = (nt >> 2) * kPlane84 + (2 * (nt & 3) + (col >> 3)) * kD84 + (col & 7);
  }
  Batch<kPlane84, uint32_t> pre_x;           // the NEXT observation's bytes and dz1 (see Batch)
  Batch<32 * kM84 / 4, float4> pre_z;
  if ((int)blockIdx.x < n_obs) {
    pre_x.load(reinterpret_cast<const uint32_t*>(obs + (size_t)blockIdx.x * 4 * kPlane84), tid);
    pre_z.load(reinterpret_cast<const float4*>(dz1 + (size_t)blockIdx.x * 32 * kM84), tid);
  }
#pragma unroll 1
  for (int n = blockIdx.x; n < n_obs; n += gridDim.x) {
    __syncthreads();
    pre_x.each(tid, [&](int wi, uint32_t v) {
      const int i = wi * 4;
      const int c = i / kPlane84, r = i - c * kPlane84, y = r / kD84, x = r - y * kD84;
      if (y < kD84 - 1) {
        uint8_t* d = tile + c * kPlane84 + (y + 1) * kD84 + (x + 1);
        d[0] = (uint8_t)v; d[1] = (uint8_t)(v >> 8); d[2] = (uint8_t)(v >> 16);
        if (x + 4 < kD84) d[3] = (uint8_t)(v >> 24);
      }
    });
    pre_z.each(tid, [&](int i, float4 v) { reinterpret_cast<float4*>(z1s)[i] = v; });
    if (n + (int)gridDim.x < n_obs) {
      const size_t nn = (size_t)n + gridDim.x;
      pre_x.load(reinterpret_cast<const uint32_t*>(obs + nn * 4 * kPlane84), tid);
      pre_z.load(reinterpret_cast<const float4*>(dz1 + nn * 32 * kM84), tid);
    }
    __syncthreads();
    // The operands of k-step ks + 1 are read from LDS BEFORE the MFMAs of k-step ks are issued (round 4; rounds 1-3
    // read a byte, then the table entry it selects, then issued the eight MFMAs that wait for both: two LDS round
    // trips per 256 clocks of the matrix pipe, 25 % of the f32 peak), and the byte becomes (float)u / 255.0f in
    // registers (byte_over_255: bit-identical to the division for all 256 bytes) instead of through the table.
    int ox = q, poff = q * 4;                               // p = 4 ks + q: (oy, ox) = (0, q); tile offset (4 oy) * 84 + 4 ox
    const float* za = z1s + col * kM84 + q;
    float a0 = za[0], a1v = za[16 * kM84];
    uint32_t ub[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ub[j] = tile[boff[j] + poff];
#pragma clang loop unroll(disable)
    for (int ks = 0; ks < 100; ++ks) {
      ox += 4;
      const bool wrap = ox >= kO84;
      ox -= wrap ? kO84 : 0;
      poff += wrap ? 16 + (4 * kD84 - 4 * kO84) : 16;       // four rows down, 80 columns back
      const bool more = ks < 99;
      const int zn = more ? (ks + 1) * 4 : 0, pn = more ? poff : 0;   // (the last step re-reads valid addresses)
      const float na0 = za[zn], na1 = za[16 * kM84 + zn];
      uint32_t nb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) nb[j] = tile[boff[j] + pn];
      __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise sinks the reads of a0 / a1v to their use in the next iteration)
      dba[0] += a0;
      dba[1] += a1v;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float b = byte_over_255(ub[j]);
        accw[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, accw[0][j], 0, 0, 0);
        accw[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v, b, accw[1][j], 0, 0, 0);
      }
      a0 = na0;
      a1v = na1;
#pragma unroll
      for (int j = 0; j < 4; ++j) ub[j] = nb[j];
    }
  }
  float* P = partial + (size_t)blockIdx.x * kPart1;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) P[(16 * mt + 4 * q + r) * 256 + 16 * (wave + 4 * j) + col] = accw[mt][j][r];
  __syncthreads();
  if (wave == 0) { red[(0 * 4 + q) * 16 + col] = dba[0]; red[(1 * 4 + q) * 16 + col] = dba[1]; }
  __syncthreads();
  if (tid < 32) {
    const int mt = tid >> 4, c = tid & 15;
    P[32 * 256 + tid] = red[(mt * 4 + 0) * 16 + c] + red[(mt * 4 + 1) * 16 + c] + red[(mt * 4 + 2) * 16 + c] +
                        red[(mt * 4 + 3) * 16 + c];
  }
}

// out[j] = sum over parts of partial[part][j], fixed order (deterministic)
__global__ __launch_bounds__(256) void partial_sum_kernel(const float* __restrict__ partial, int n_parts, int len,
                                                          float* __restrict__ out) {
  __shared__ float red[256];
  const int j = blockIdx.x * 16 + (threadIdx.x & 15);
  const float s = partial_sum16(partial, n_parts, len, j, red);
  if (threadIdx.x < 16 && j < len) out[j] = s;
}

}  // namespace parlhip

using namespace parlhip;

static int launch_conv12(const uint8_t* obs, const RingObs& ro, const float* w1, const float* b1, const float* w2,
                         const float* b2, float* out, int n_obs, const float* packed, hipStream_t stream,
                         float* a1_out = nullptr) {
  const size_t lds_bytes = kLdsFloats * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    const void* fns[5] = {(const void*)conv12_u8_mfma_kernel<false, false, false>, (const void*)conv12_u8_mfma_kernel<false, true, false>,
                          (const void*)conv12_u8_mfma_kernel<true, false, false>, (const void*)conv12_u8_mfma_kernel<true, true, false>,
                          (const void*)conv12_u8_mfma_kernel<false, true, true>};
    for (const void* f : fns) {
      int rc = check(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
      if (rc) return rc;
    }
    attr_set = true;
  }
  // The learner's launches (SAVE forward, backward) run ONE workgroup per CU, the actors' two.  Alone that costs the
  // learner 7 % (one 1000-row update 0.250 -> 0.269 ms: a single workgroup has nobody to hide its fills behind), but
  // its conv kernels are what stretches the emulator's waves beside them and one learner wave per SIMD
  // instead of two stretches them less: pipeline 6.26-6.30 -> 6.42-6.46 M frames/s (128 / 192 / 384 workgroups:
  // 6.12-6.15 / 6.36 / 6.12-6.14 M; round 6, same box, two runs each).
#ifndef PARLHIP_C12_LEARNER_GRID
#define PARLHIP_C12_LEARNER_GRID kNumCU
#endif
  const int max_grid = a1_out ? PARLHIP_C12_LEARNER_GRID : 2 * kNumCU;
  const int grid = n_obs < max_grid ? n_obs : max_grid;  // 71 KB of LDS: two workgroups per CU
#define PARLHIP_C12(R, P, S, O, RO) \
  conv12_u8_mfma_kernel<R, P, S><<<grid, 256, lds_bytes, stream>>>(O, RO, w1, b1, w2, b2, out, n_obs, packed, a1_out)
  if (a1_out) {
    if (ro.ring || !packed) return PARLHIP_EINVAL;
    PARLHIP_C12(false, true, true, obs, RingObs{});
  } else if (ro.ring) {
    if (packed) PARLHIP_C12(true, true, false, nullptr, ro);
    else PARLHIP_C12(true, false, false, nullptr, ro);
  } else {
    if (packed) PARLHIP_C12(false, true, false, obs, RingObs{});
    else PARLHIP_C12(false, false, false, obs, RingObs{});
  }
#undef PARLHIP_C12
  return check_launch();
}

PARLHIP_EXPORT int parlhip_atari42_conv12_u8_f32(const uint8_t* obs, const float* w1, const float* b1,
                                                 const float* w2, const float* b2, float* out, int n_obs,
                                                 parlhip_stream_t stream) {
  if (n_obs < 0) return PARLHIP_EINVAL;
  if (n_obs == 0) return PARLHIP_OK;
  if (!obs || !w1 || !b1 || !w2 || !b2 || !out) return PARLHIP_EINVAL;
  return launch_conv12(obs, RingObs{}, w1, b1, w2, b2, out, n_obs, nullptr, (hipStream_t)stream);
}

PARLHIP_EXPORT int parlhip_atari42_conv12_ring_u8_f32(const uint8_t* ring, const uint8_t* since, int num_slots, int E,
                                                      int slot, const float* w1, const float* b1, const float* w2,
                                                      const float* b2, float* out, parlhip_stream_t stream) {
  if (E < 0 || num_slots < 4 || slot < 0 || slot >= num_slots) return PARLHIP_EINVAL;
  if (E == 0) return PARLHIP_OK;
  if (!ring || !since || !w1 || !b1 || !w2 || !b2 || !out) return PARLHIP_EINVAL;
  if ((uintptr_t)ring & 3u) return PARLHIP_EINVAL;
  return launch_conv12(nullptr, RingObs{ring, since, num_slots, E, slot}, w1, b1, w2, b2, out, E, nullptr,
                       (hipStream_t)stream);
}

PARLHIP_EXPORT size_t parlhip_atari42_conv12_weights_bytes(void) { return (size_t)kPackedFloats * sizeof(float); }

PARLHIP_EXPORT int parlhip_atari42_conv12_weights_f32(const float* w1, const float* w2, float* packed_out,
                                                      parlhip_stream_t stream) {
  if (!w1 || !w2 || !packed_out) return PARLHIP_EINVAL;
  if (reinterpret_cast<uintptr_t>(packed_out) & 15) return PARLHIP_EINVAL;
  conv12_weights_pack_kernel<<<(kPackedFloats + 255) / 256, 256, 0, (hipStream_t)stream>>>(w1, w2, packed_out);
  return check_launch();
}

PARLHIP_EXPORT int parlhip_atari42_conv12_packed_u8_f32(const uint8_t* obs, const float* packed, const float* b1,
                                                        const float* b2, float* out, int n_obs,
                                                        parlhip_stream_t stream) {
  if (n_obs < 0) return PARLHIP_EINVAL;
  if (n_obs == 0) return PARLHIP_OK;
  if (!obs || !packed || !b1 || !b2 || !out) return PARLHIP_EINVAL;
  if (reinterpret_cast<uintptr_t>(packed) & 15) return PARLHIP_EINVAL;
  return launch_conv12(obs, RingObs{}, nullptr, b1, nullptr, b2, out, n_obs, packed, (hipStream_t)stream);
}

PARLHIP_EXPORT size_t parlhip_atari42_conv12_a1_bytes(int n_obs) {
  return n_obs <= 0 ? 0 : (size_t)n_obs * kA1Row * sizeof(float);
}

PARLHIP_EXPORT int parlhip_atari42_conv12_packed_save_u8_f32(const uint8_t* obs, const float* packed, const float* b1,
                                                             const float* b2, float* out, float* a1_out, int n_obs,
                                                             parlhip_stream_t stream) {
  if (n_obs < 0) return PARLHIP_EINVAL;
  if (n_obs == 0) return PARLHIP_OK;
  if (!obs || !packed || !b1 || !b2 || !out || !a1_out) return PARLHIP_EINVAL;
  if ((reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(a1_out)) & 15) return PARLHIP_EINVAL;
  return launch_conv12(obs, RingObs{}, nullptr, b1, nullptr, b2, out, n_obs, packed, (hipStream_t)stream, a1_out);
}

PARLHIP_EXPORT int parlhip_atari42_conv12_ring_packed_u8_f32(const uint8_t* ring, const uint8_t* since, int num_slots,
                                                             int E, int slot, const float* packed, const float* b1,
                                                             const float* b2, float* out, parlhip_stream_t stream) {
  if (E < 0 || num_slots < 4 || slot < 0 || slot >= num_slots) return PARLHIP_EINVAL;
  if (E == 0) return PARLHIP_OK;
  if (!ring || !since || !packed || !b1 || !b2 || !out) return PARLHIP_EINVAL;
  if (((uintptr_t)ring & 3u) || (reinterpret_cast<uintptr_t>(packed) & 15)) return PARLHIP_EINVAL;
  return launch_conv12(nullptr, RingObs{ring, since, num_slots, E, slot}, nullptr, b1, nullptr, b2, out, E, packed,
                       (hipStream_t)stream);
}

static int launch_conv1_84(const uint8_t* obs, const RingObs& ro, const float* w, const float* b1, float* out, int n_obs,
                           bool packed, hipStream_t stream) {
  static bool attr_set = false;
  const size_t lds_bytes = kLds84u8Bytes;
  if (!attr_set) {
    const void* fns[4] = {(const void*)conv1_84_u8_mfma_kernel<false, false>, (const void*)conv1_84_u8_mfma_kernel<false, true>,
                          (const void*)conv1_84_u8_mfma_kernel<true, false>, (const void*)conv1_84_u8_mfma_kernel<true, true>};
    for (const void* f : fns) {
      int rc = check(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
      if (rc) return rc;
    }
    attr_set = true;
  }
  const int grid = n_obs < c84_per_cu(n_obs) * kNumCU ? n_obs : c84_per_cu(n_obs) * kNumCU;  // 28 KB of LDS, <= 256 VGPRs: two workgroups per CU
#define PARLHIP_C184(R, P, O, RO) conv1_84_u8_mfma_kernel<R, P><<<grid, 256, lds_bytes, stream>>>(O, RO, w, b1, out, n_obs)
  if (ro.ring) {
    if (packed) PARLHIP_C184(true, true, nullptr, ro);
    else PARLHIP_C184(true, false, nullptr, ro);
  } else {
    if (packed) PARLHIP_C184(false, true, obs, RingObs{});
    else PARLHIP_C184(false, false, obs, RingObs{});
  }
#undef PARLHIP_C184
  return check_launch();
}

static int conv1_84_args(const void* in, const float* w1, const float* b1, const float* out, int n_obs) {
  if (n_obs < 0) return PARLHIP_EINVAL;
  if (n_obs == 0) return 1;
  if (!in || !w1 || !b1 || !out) return PARLHIP_EINVAL;
  // 4-byte input loads / 16-byte output stores (torch allocations are 256 B aligned; a view
  // starting at an observation boundary keeps both: 28,224 and 51,200 are multiples of 16)
  if (((uintptr_t)in & 3u) || ((uintptr_t)out & 15u)) return PARLHIP_EINVAL;
  return PARLHIP_OK;
}

PARLHIP_EXPORT int parlhip_atari84_conv1_u8_f32(const uint8_t* obs, const float* w1, const float* b1,
                                                float* out, int n_obs, parlhip_stream_t stream) {
  const int rc = conv1_84_args(obs, w1, b1, out, n_obs);
  if (rc) return rc < 0 ? rc : PARLHIP_OK;
  return launch_conv1_84(obs, RingObs{}, w1, b1, out, n_obs, false, (hipStream_t)stream);
}

PARLHIP_EXPORT int parlhip_atari84_conv1_packed_u8_f32(const uint8_t* obs, const float* wt1, const float* b1,
                                                       float* out, int n_obs, parlhip_stream_t stream) {
  const int rc = conv1_84_args(obs, wt1, b1, out, n_obs);
  if (rc) return rc < 0 ? rc : PARLHIP_OK;
  return launch_conv1_84(obs, RingObs{}, wt1, b1, out, n_obs, true, (hipStream_t)stream);
}

PARLHIP_EXPORT int parlhip_atari84_conv1_ring_u8_f32(const uint8_t* ring, const uint8_t* since, int num_slots, int E,
                                                     int slot, const float* w1, const float* b1, float* out,
                                                     parlhip_stream_t stream) {
  if (num_slots < 4 || slot < 0 || slot >= num_slots) return PARLHIP_EINVAL;
  const int rc = conv1_84_args(ring, w1, b1, out, E);
  if (rc) return rc < 0 ? rc : PARLHIP_OK;
  if (!since) return PARLHIP_EINVAL;
  return launch_conv1_84(nullptr, RingObs{ring, since, num_slots, E, slot}, w1, b1, out, E, false, (hipStream_t)stream);
}

PARLHIP_EXPORT int parlhip_atari84_conv1_ring_packed_u8_f32(const uint8_t* ring, const uint8_t* since, int num_slots,
                                                            int E, int slot, const float* wt1, const float* b1,
                                                            float* out, parlhip_stream_t stream) {
  if (num_slots < 4 || slot < 0 || slot >= num_slots) return PARLHIP_EINVAL;
  const int rc = conv1_84_args(ring, wt1, b1, out, E);
  if (rc) return rc < 0 ? rc : PARLHIP_OK;
  if (!since) return PARLHIP_EINVAL;
  return launch_conv1_84(nullptr, RingObs{ring, since, num_slots, E, slot}, wt1, b1, out, E, true, (hipStream_t)stream);
}

#ifndef PARLHIP_C12_LEARNER_GRID
#define PARLHIP_C12_LEARNER_GRID kNumCU
#endif
#ifndef PARLHIP_C12_LEARNER_BWD_GRID
#define PARLHIP_C12_LEARNER_BWD_GRID PARLHIP_C12_LEARNER_GRID
#endif
static int conv12_bwd_grid(int n_obs) { return n_obs < PARLHIP_C12_LEARNER_BWD_GRID ? n_obs : PARLHIP_C12_LEARNER_BWD_GRID; }  // (69 KB of LDS: two would fit)

PARLHIP_EXPORT size_t parlhip_atari42_conv12_bwd_workspace_bytes(int n_obs) {
  return n_obs <= 0 ? 0 : (size_t)conv12_bwd_grid(n_obs) * kBwdPartial * sizeof(float);
}

static int launch_conv12_bwd(const uint8_t* obs, const float* w1_or_packed, const float* b1, const float* w2,
                             const float* a2, const float* dy, int n_obs, float* workspace, float* dw1, float* db1,
                             float* dw2, float* db2, bool packed, hipStream_t s, const float* a1 = nullptr) {
  if (n_obs < 0) return PARLHIP_EINVAL;
  if (!dw1 || !db1 || !dw2 || !db2) return PARLHIP_EINVAL;
  if (n_obs == 0) {
    int rc = check(hipMemsetAsync(dw1, 0, 1024 * 4, s));
    if (!rc) rc = check(hipMemsetAsync(db1, 0, 16 * 4, s));
    if (!rc) rc = check(hipMemsetAsync(dw2, 0, 8192 * 4, s));
    if (!rc) rc = check(hipMemsetAsync(db2, 0, 32 * 4, s));
    return rc;
  }
  if (!obs || !w1_or_packed || !b1 || (!packed && !w2) || !a2 || !dy || !workspace) return PARLHIP_EINVAL;
  if (packed && (reinterpret_cast<uintptr_t>(w1_or_packed) & 15)) return PARLHIP_EINVAL;
  static bool attr_set = false;
  const size_t lds_bytes = kLdsBwdFloats * sizeof(float);
  if (a1 && (!packed || (reinterpret_cast<uintptr_t>(a1) & 15))) return PARLHIP_EINVAL;
  if (!attr_set) {
    const void* fns[3] = {(const void*)conv12_bwd_u8_mfma_kernel<false, false>, (const void*)conv12_bwd_u8_mfma_kernel<true, false>,
                          (const void*)conv12_bwd_u8_mfma_kernel<true, true>};
    for (const void* f : fns) {
      int rc = check(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
      if (rc) return rc;
    }
    attr_set = true;
  }
  const int grid = conv12_bwd_grid(n_obs);
  if (a1) conv12_bwd_u8_mfma_kernel<true, true><<<grid, 256, lds_bytes, s>>>(obs, w1_or_packed, b1, nullptr, a2, dy, workspace, n_obs, a1);
  else if (packed) conv12_bwd_u8_mfma_kernel<true, false><<<grid, 256, lds_bytes, s>>>(obs, w1_or_packed, b1, nullptr, a2, dy, workspace, n_obs, nullptr);
  else conv12_bwd_u8_mfma_kernel<false, false><<<grid, 256, lds_bytes, s>>>(obs, w1_or_packed, b1, w2, a2, dy, workspace, n_obs, nullptr);
  int rc = check_launch();
  if (rc) return rc;
  conv12_bwd_reduce_kernel<<<(kBwdPartial + 15) / 16, 256, 0, s>>>(workspace, grid, dw1, db1, dw2, db2);
  return check_launch();
}

PARLHIP_EXPORT int parlhip_atari42_conv12_bwd_f32(const uint8_t* obs, const float* w1, const float* b1,
                                                  const float* w2, const float* a2, const float* dy, int n_obs,
                                                  float* workspace, float* dw1, float* db1, float* dw2, float* db2,
                                                  parlhip_stream_t stream) {
  return launch_conv12_bwd(obs, w1, b1, w2, a2, dy, n_obs, workspace, dw1, db1, dw2, db2, false, (hipStream_t)stream);
}

PARLHIP_EXPORT int parlhip_atari42_conv12_bwd_packed_f32(const uint8_t* obs, const float* packed, const float* b1,
                                                         const float* a2, const float* dy, int n_obs,
                                                         float* workspace, float* dw1, float* db1, float* dw2,
                                                         float* db2, parlhip_stream_t stream) {
  return launch_conv12_bwd(obs, packed, b1, nullptr, a2, dy, n_obs, workspace, dw1, db1, dw2, db2, true,
                           (hipStream_t)stream);
}

PARLHIP_EXPORT int parlhip_atari42_conv12_bwd_saved_f32(const uint8_t* obs, const float* packed, const float* b1,
                                                        const float* a1, const float* a2, const float* dy, int n_obs,
                                                        float* workspace, float* dw1, float* db1, float* dw2,
                                                        float* db2, parlhip_stream_t stream) {
  if (n_obs > 0 && !a1) return PARLHIP_EINVAL;
  return launch_conv12_bwd(obs, packed, b1, nullptr, a2, dy, n_obs, workspace, dw1, db1, dw2, db2, true,
                           (hipStream_t)stream, a1);
}

PARLHIP_EXPORT int parlhip_atari84_conv23_f32(const float* a1, const float* wt2, const float* b2, const float* wt3,
                                              const float* b3, float* a2_out, float* a3_out, int n_obs,
                                              parlhip_stream_t stream) {
  if (n_obs < 0) return PARLHIP_EINVAL;
  if (n_obs == 0) return PARLHIP_OK;
  if (!a1 || !wt2 || !b2 || !wt3 || !b3 || !a3_out) return PARLHIP_EINVAL;
  if ((uintptr_t)a1 & 15u) return PARLHIP_EINVAL;   // 16-byte loads of the conv1 activation
  static bool attr_set = false;
  const size_t lds_bytes = kLds23Floats * sizeof(float);
  if (!attr_set) {
    int rc = check(hipFuncSetAttribute((const void*)conv23_84_mfma_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    if (rc) return rc;
    attr_set = true;
  }
  // 49 KB of LDS: up to three workgroups per CU (two are launched: more only shortens the streamed-weight reuse),
  // one beside a learner workgroup
  const int grid = n_obs < c84_per_cu(n_obs) * kNumCU ? n_obs : c84_per_cu(n_obs) * kNumCU;
  conv23_84_mfma_kernel<<<grid, 256, lds_bytes, (hipStream_t)stream>>>(a1, wt2, b2, wt3, b3, a2_out, a3_out, n_obs);
  return check_launch();
}

static int bwd84_grid(int n_obs) { return n_obs < kNumCU ? n_obs : kNumCU; }

PARLHIP_EXPORT size_t parlhip_atari84_conv3_bwd_workspace_bytes(int n_obs) {
  return n_obs <= 0 ? 0 : (size_t)bwd84_grid(n_obs) * kPart3 * sizeof(float);
}

PARLHIP_EXPORT int parlhip_atari84_conv3_bwd_f32(const float* a2, const float* a3, const float* dy3, const float* wt3b,
                                                 int n_obs, float* workspace, float* dz2, float* dw3_db3,
                                                 parlhip_stream_t stream) {
  if (n_obs <= 0) return n_obs < 0 ? PARLHIP_EINVAL : PARLHIP_OK;
  if (!a2 || !a3 || !dy3 || !wt3b || !workspace || !dz2 || !dw3_db3) return PARLHIP_EINVAL;
  if ((uintptr_t)a2 & 15u) return PARLHIP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  static bool attr_set = false;
  const size_t lds_bytes = kLds3bFloats * sizeof(float);
  if (!attr_set) {
    int rc = check(hipFuncSetAttribute((const void*)conv3_84_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds_bytes));
    if (rc) return rc;
    attr_set = true;
  }
  const int grid = bwd84_grid(n_obs);
  conv3_84_bwd_kernel<<<grid, 256, lds_bytes, s>>>(a2, a3, dy3, wt3b, dz2, workspace, n_obs);
  int rc = check_launch();
  if (rc) return rc;
  partial_sum_kernel<<<(kPart3 + 15) / 16, 256, 0, s>>>(workspace, grid, kPart3, dw3_db3);
  return check_launch();
}

PARLHIP_EXPORT size_t parlhip_atari84_conv2_bwd_workspace_bytes(int n_obs) {
  return n_obs <= 0 ? 0 : (size_t)bwd84_grid(n_obs) * kPart2 * sizeof(float);
}

PARLHIP_EXPORT int parlhip_atari84_conv2_bwd_f32(const float* a1, const float* dz2, const float* wt2b, int n_obs,
                                                 float* workspace, float* dz1, float* dw2_db2, parlhip_stream_t stream) {
  if (n_obs <= 0) return n_obs < 0 ? PARLHIP_EINVAL : PARLHIP_OK;
  if (!a1 || !dz2 || !wt2b || !workspace || !dz1 || !dw2_db2) return PARLHIP_EINVAL;
  if (((uintptr_t)a1 | (uintptr_t)dz2) & 15u) return PARLHIP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  static bool attr_set = false;
  const size_t lds_bytes = kLds2bFloats * sizeof(float);
  if (!attr_set) {
    int rc = check(hipFuncSetAttribute((const void*)conv2_84_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds_bytes));
    if (rc) return rc;
    attr_set = true;
  }
  const int grid = bwd84_grid(n_obs);
  conv2_84_bwd_kernel<<<grid, 256, lds_bytes, s>>>(a1, dz2, wt2b, dz1, workspace, n_obs);
  int rc = check_launch();
  if (rc) return rc;
  partial_sum_kernel<<<(kPart2 + 15) / 16, 256, 0, s>>>(workspace, grid, kPart2, dw2_db2);
  return check_launch();
}

// conv1_84_bwd_kernel: 80.7 KB of LDS and <= 256 VGPRs — two workgroups per CU, each one's fill and store phases under
// the other's MFMAs (conv3 / conv2: 276+ VGPRs of accumulators, one wave per SIMD whatever the grid)
static int bwd84_conv1_grid(int n_obs) { return n_obs < c84_per_cu(n_obs) * kNumCU ? n_obs : c84_per_cu(n_obs) * kNumCU; }

PARLHIP_EXPORT size_t parlhip_atari84_conv1_bwd_workspace_bytes(int n_obs) {
  return n_obs <= 0 ? 0 : (size_t)bwd84_conv1_grid(n_obs) * kPart1 * sizeof(float);
}

PARLHIP_EXPORT int parlhip_atari84_conv1_bwd_f32(const uint8_t* obs, const float* dz1, int n_obs, float* workspace,
                                                 float* dw1_db1, parlhip_stream_t stream) {
  if (n_obs <= 0) return n_obs < 0 ? PARLHIP_EINVAL : PARLHIP_OK;
  if (!obs || !dz1 || !workspace || !dw1_db1) return PARLHIP_EINVAL;
  if (((uintptr_t)obs & 3u) || ((uintptr_t)dz1 & 15u)) return PARLHIP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  static bool attr_set = false;
  const size_t lds_bytes = kLds1bFloats * sizeof(float);
  if (!attr_set) {
    int rc = check(hipFuncSetAttribute((const void*)conv1_84_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds_bytes));
    if (rc) return rc;
    attr_set = true;
  }
  const int grid = bwd84_conv1_grid(n_obs);
  conv1_84_bwd_kernel<<<grid, 256, lds_bytes, s>>>(obs, dz1, workspace, n_obs);
  int rc = check_launch();
  if (rc) return rc;
  partial_sum_kernel<<<(kPart1 + 15) / 16, 256, 0, s>>>(workspace, grid, kPart1, dw1_db1);
  return check_launch();
}

#ifdef PARLHIP_CONV_REGIONS
PARLHIP_EXPORT int parlhip_debug_conv_regions(unsigned long long* host, int reset) {
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(parlhip::g_conv_regions), 128) != hipSuccess) return PARLHIP_EINVAL;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(parlhip::g_conv_regions), z, 128) != hipSuccess) return PARLHIP_EINVAL;
  }
  return PARLHIP_OK;
}
#endif
