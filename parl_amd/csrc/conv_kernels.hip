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

constexpr int kD = 42, kP1 = 44;          // input, zero-padded input (pad 1, +1 slack column/row)
constexpr int kO1 = 21, kC1 = 16;         // conv1 output size / channels
constexpr int kP2 = 25;                   // zero-padded conv1 output (pad 2)
constexpr int kO2 = 11, kC2 = 32;         // conv2 output size / channels
constexpr int kM1 = kO1 * kO1, kM2 = kO2 * kO2;
constexpr int kK1 = 64, kK2 = 256;
constexpr int kLdsIn = 4 * kP1 * kP1;     // 7744 floats
constexpr int kLdsC1 = kC1 * kP2 * kP2;   // 10000
constexpr int kLdsFloats = kLdsIn + kLdsC1;  // 17,744 floats = 70,976 B: two workgroups per CU

// Both weight matrices live in registers (B operands: 16 + 128 VGPRs per lane, loaded once per
// workgroup), so an MFMA needs one LDS gather (conv1) or half of one (conv2: both N-tiles reuse
// the A value); outputs leave the accumulators straight for HBM; the zero borders of the two LDS
// tiles are written once and never touched again (only interiors are rewritten per observation).
__global__ __launch_bounds__(256) void conv12_u8_mfma_kernel(
    const uint8_t* __restrict__ obs, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ out, int n_obs) {
  extern __shared__ float lds[];
  float* in_pad = lds;                  // [4][44][44]
  float* c1_pad = in_pad + kLdsIn;      // [16][25][25]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, col = lane & 15;
  // B[k][n] = w[n][k]: lane (q, col) holds k = 4*ks + q, n = col (+16 for the second N-tile)
  float bw1[16], bw2[64][2];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) bw1[ks] = w1[col * kK1 + ks * 4 + q];
#pragma unroll
  for (int ks = 0; ks < 64; ++ks) {
    bw2[ks][0] = w2[col * kK2 + ks * 4 + q];
    bw2[ks][1] = w2[(16 + col) * kK2 + ks * 4 + q];
  }
  const float bias1 = b1[col], bias20 = b2[col], bias21 = b2[16 + col];
  for (int i = tid; i < kLdsFloats; i += 256) lds[i] = 0.0f;
  for (int n = blockIdx.x; n < n_obs; n += gridDim.x) {
    __syncthreads();  // borders zeroed / the previous observation's conv2 gathers are done
    // ---- obs u8 -> padded float input (x / 255, the division as in the reference) ----
    const uint8_t* src = obs + (size_t)n * 4 * kD * kD;
    if ((reinterpret_cast<uintptr_t>(src) & 3) == 0) {  // wave-uniform: 7056 B per observation
      const uint32_t* src32 = reinterpret_cast<const uint32_t*>(src);
      for (int wi = tid; wi < kD * kD; wi += 256) {   // 4 * 1764 bytes = 1764 words
        const uint32_t v = src32[wi];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = wi * 4 + j;
          const int c = i / (kD * kD), r = i - c * kD * kD, y = r / kD, x = r - y * kD;
          in_pad[c * kP1 * kP1 + (y + 1) * kP1 + (x + 1)] = (float)((v >> (8 * j)) & 255u) / 255.0f;
        }
      }
    } else {
      for (int i = tid; i < 4 * kD * kD; i += 256) {
        const int c = i / (kD * kD), r = i - c * kD * kD, y = r / kD, x = r - y * kD;
        in_pad[c * kP1 * kP1 + (y + 1) * kP1 + (x + 1)] = (float)src[i] / 255.0f;
      }
    }
    __syncthreads();
    // ---- conv1: 28 M-tiles of 16 positions, 7 per wave ----
    for (int t = wave; t < 28; t += 4) {
      int m = t * 16 + col;
      m = m < kM1 ? m : kM1 - 1;
      const int oy = m / kO1, ox = m - oy * kO1;
      const float* a_base = in_pad + (2 * oy) * kP1 + 2 * ox + q;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const float a = a_base[(ks >> 2) * kP1 * kP1 + (ks & 3) * kP1];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw1[ks], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mo = t * 16 + q * 4 + r;  // D row
        if (mo < kM1) {
          const int y = mo / kO1, x = mo - y * kO1;
          const float v = acc[r] + bias1;
          c1_pad[col * kP2 * kP2 + (y + 2) * kP2 + (x + 2)] = v > 0.f ? v : 0.f;
        }
      }
    }
    __syncthreads();
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
        const float a = a_base[(ks >> 2) * kP2 * kP2 + (ks & 3) * kP2];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw2[ks][0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw2[ks][1], acc1, 0, 0, 0);
      }
      // D: column = channel, rows 4q..4q+3 = 4 consecutive positions of the [32][121] row
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mo = mt * 16 + q * 4 + r;
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
// conv1 of the A2C Atari network (the 84x84 -> 20x20 contraction) on the f32 matrix cores.
//
// Reference: examples/A2C/atari_model.py:21-104 (AtariModel): obs / 255, conv1 4->32 k8 s4 p1
// + ReLU (84x84 -> 20x20).  Per observation an implicit GEMM [400 positions x 256] x [256 x 32]
// (6.55 MFLOP), k = c*64 + kh*8 + kw (the order of weight.flatten(1)).
//
// One workgroup (4 wavefronts, one per SIMD) per observation, grid-stride over observations:
//   * the u8 stack is read once with 4-byte loads (28,224 B), divided by 255 and laid out in LDS
//     shifted by the padding: tile[c][py][px] = obs[c][py-1][px-1] / 255, row 0 / column 0 = 0.
//     (pad 1 with floor((84+2-8)/4)+1 = 20 outputs: padded rows/columns 84 and 85 are never read,
//     so the tile is 4 x 84 x 84 floats = 112,896 B);
//   * the whole B operand (the 32 KB weight matrix) lives in registers: 64 k-steps x 2 N-tiles =
//     128 VGPRs per lane, loaded once per workgroup (a single wave per SIMD has 512 VGPRs);
//   * each wave owns M-tiles (16 output positions) and issues 2 MFMAs per A gather (both N-tiles);
//   * D (col = channel, rows = 4 consecutive positions) + bias + ReLU goes straight to HBM as one
//     16-byte store per lane in NCHW order — no staging, no im2col matrix in HBM (the GEMM-lowered
//     conv writes and re-reads 410 KB of patches per observation).
// Algorithmic bytes per observation: 28,224 read + 51,200 written.
// ----------------------------------------------------------------------------------------
constexpr int kD84 = 84;                 // input size = padded-tile size (see above)
constexpr int kO84 = 20, kC84 = 32;      // conv1 output size / channels
constexpr int kM84 = kO84 * kO84;        // 400 positions = 25 M-tiles of 16
constexpr int kK84 = 4 * 8 * 8;          // 256
constexpr int kPlane84 = kD84 * kD84;    // 7056
constexpr int kLds84Floats = 4 * kPlane84;

__global__ __launch_bounds__(256) void conv1_84_u8_mfma_kernel(
    const uint8_t* __restrict__ obs, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ out, int n_obs) {
  extern __shared__ float lds[];  // [4][84][84], shifted by the padding
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, col = lane & 15;
  // B[k][n] = w[n][k]: lane (q, col) holds k = 4*ks + q, n = 16*nt + col
  float breg[64][2];
#pragma unroll
  for (int ks = 0; ks < 64; ++ks) {
    breg[ks][0] = w[col * kK84 + ks * 4 + q];
    breg[ks][1] = w[(16 + col) * kK84 + ks * 4 + q];
  }
  const float bias0 = bias[col], bias1 = bias[16 + col];
  // the padding: row 0 and column 0 of every channel plane (never overwritten below)
  for (int i = tid; i < 4 * kD84; i += 256) {
    const int c = i / kD84, p = i - c * kD84;
    lds[c * kPlane84 + p] = 0.0f;
    lds[c * kPlane84 + p * kD84] = 0.0f;
  }
  for (int n = blockIdx.x; n < n_obs; n += gridDim.x) {
    __syncthreads();  // the previous observation's gathers are done before the tile is rewritten
    const uint32_t* src = reinterpret_cast<const uint32_t*>(obs + (size_t)n * 4 * kPlane84);
    for (int wi = tid; wi < kPlane84; wi += 256) {  // 4 * 7056 bytes = 7056 words; 84 % 4 == 0
      const uint32_t v = src[wi];
      const int i = wi * 4;
      const int c = i / kPlane84, r = i - c * kPlane84, y = r / kD84, x = r - y * kD84;
      if (y < kD84 - 1) {  // input row 83 lies outside every window
        float* d = lds + c * kPlane84 + (y + 1) * kD84 + (x + 1);
        d[0] = (float)(v & 255u) / 255.0f;
        d[1] = (float)((v >> 8) & 255u) / 255.0f;
        d[2] = (float)((v >> 16) & 255u) / 255.0f;
        if (x + 4 < kD84) d[3] = (float)(v >> 24) / 255.0f;  // input column 83 likewise
      }
    }
    __syncthreads();
    float* dst = out + (size_t)n * kC84 * kM84;
    for (int mt = wave; mt < kM84 / 16; mt += 4) {
      const int m = mt * 16 + col;
      const int oy = m / kO84, ox = m - oy * kO84;
      const float* a_base = lds + (4 * oy) * kD84 + 4 * ox + q;
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 64; ++ks) {
        // k = 4*ks + q = c*64 + kh*8 + kw  ->  c = ks>>4, kh = (ks>>1)&7, kw = (ks&1)*4 + q
        const float a = a_base[(ks >> 4) * kPlane84 + ((ks >> 1) & 7) * kD84 + (ks & 1) * 4];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, breg[ks][0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, breg[ks][1], acc1, 0, 0, 0);
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

}  // namespace parlhip

using namespace parlhip;

PARLHIP_EXPORT int parlhip_atari42_conv12_u8_f32(const uint8_t* obs, const float* w1, const float* b1,
                                                 const float* w2, const float* b2, float* out, int n_obs,
                                                 parlhip_stream_t stream) {
  if (n_obs < 0) return PARLHIP_EINVAL;
  if (n_obs == 0) return PARLHIP_OK;
  if (!obs || !w1 || !b1 || !w2 || !b2 || !out) return PARLHIP_EINVAL;
  static bool attr_set = false;
  const size_t lds_bytes = kLdsFloats * sizeof(float);
  if (!attr_set) {
    int rc = check(hipFuncSetAttribute((const void*)conv12_u8_mfma_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    if (rc) return rc;
    attr_set = true;
  }
  const int grid = n_obs < 2 * kNumCU ? n_obs : 2 * kNumCU;  // 71 KB of LDS: two workgroups per CU
  conv12_u8_mfma_kernel<<<grid, 256, lds_bytes, (hipStream_t)stream>>>(obs, w1, b1, w2, b2, out, n_obs);
  return check_launch();
}

PARLHIP_EXPORT int parlhip_atari84_conv1_u8_f32(const uint8_t* obs, const float* w1, const float* b1,
                                                float* out, int n_obs, parlhip_stream_t stream) {
  if (n_obs < 0) return PARLHIP_EINVAL;
  if (n_obs == 0) return PARLHIP_OK;
  if (!obs || !w1 || !b1 || !out) return PARLHIP_EINVAL;
  // 4-byte input loads / 16-byte output stores (torch allocations are 256 B aligned; a view
  // starting at an observation boundary keeps both: 28,224 and 51,200 are multiples of 16)
  if (((uintptr_t)obs & 3u) || ((uintptr_t)out & 15u)) return PARLHIP_EINVAL;
  static bool attr_set = false;
  const size_t lds_bytes = kLds84Floats * sizeof(float);
  if (!attr_set) {
    int rc = check(hipFuncSetAttribute((const void*)conv1_84_u8_mfma_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    if (rc) return rc;
    attr_set = true;
  }
  const int grid = n_obs < kNumCU ? n_obs : kNumCU;  // 113 KB of LDS: one workgroup per CU
  conv1_84_u8_mfma_kernel<<<grid, 256, lds_bytes, (hipStream_t)stream>>>(obs, w1, b1, out, n_obs);
  return check_launch();
}
