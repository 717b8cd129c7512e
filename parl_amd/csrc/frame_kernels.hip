// frame_kernels.hip — MaxAndSkipEnv max + WarpFrame (frame_post) and FrameStack over a ring of
// single frames.  Reference: parl/env/atari_wrappers.py:239, :263-267, :270-306.
#include "common.hpp"
#include "atari_defs.hpp"
#include "frame_defs.hpp"
#include <math.h>
#include <string.h>
#include <vector>

#define DEVI __device__ __forceinline__

namespace parlhip {
namespace atari {

// ========================================================================================
// frame_post: MaxAndSkipEnv max (atari_wrappers.py:239) + WarpFrame (:263-267) as restated in
// oracle/frame_oracle.c.  One workgroup per env: the two 33,600-byte colour frames are read
// once with 16-byte loads, reduced to max-RGB gray in LDS, then area-resampled from LDS.
// Algorithmic bytes per env-step: 2*33,600 read + dim*dim written.
// ========================================================================================
// LDS of one workgroup: gray frame + palette + the tap tables, sized by the bound of area_tab
// (<= src + 2 * dst taps per axis) so that dim <= 84 fits 4 workgroups per CU (<= 40 KB each)
static inline size_t frame_post_lds_bytes(int dim) {
  return (size_t)kFrameBytes + 128 * 4 + (size_t)(kW + kH + 4 * dim) * sizeof(Tap) + 2 * (size_t)(dim + 1) * 4 + 128;  // + g1[128]
}

__global__ __launch_bounds__(512) void frame_post_kernel(
    const uint8_t* __restrict__ frames0, const uint8_t* __restrict__ frames1, int64_t in_stride,
    int fmt, const uint8_t* __restrict__ flags, uint8_t* __restrict__ out, int64_t out_stride,
    int dim, const uint8_t* __restrict__ blob, const uint8_t* __restrict__ since_prev,
    uint8_t* __restrict__ since_next, const float* __restrict__ ep_returns, const int* __restrict__ ep_lengths,
    double* __restrict__ ep_acc) {
  extern __shared__ __attribute__((aligned(16))) uint8_t fp_lds[];
  uint8_t* gray = fp_lds;                                   // [kFrameBytes] (33,600: 16-byte multiple)
  uint32_t* pal = (uint32_t*)(fp_lds + kFrameBytes);        // [128]
  Tap* s_xt = (Tap*)(pal + 128);                            // [<= kW + 2 dim]
  Tap* s_yt = s_xt + (kW + 2 * dim);                        // [<= kH + 2 dim]
  int* s_xstart = (int*)(s_yt + (kH + 2 * dim));            // [dim + 1]
  int* s_ystart = s_xstart + (dim + 1);                     // [dim + 1]
  const int e = blockIdx.x;
  if (ep_acc && threadIdx.x == 32) {  // MonitorEnv statistics of this env's step (episode_stats_kernel), same launch
    const int len = ep_lengths[e];
    if (len > 0) {
      atomicAdd(ep_acc + 0, 1.0);
      atomicAdd(ep_acc + 1, (double)ep_returns[e]);
      atomicAdd(ep_acc + 2, (double)len);
    }
  }
  if (flags && (flags[e] & 4)) {  // elastic stepping: this env delivered no observation in this launch
    if (since_next && threadIdx.x == 0) since_next[e] = 0;
    return;
  }
  if (since_next && threadIdx.x == 0) {  // FrameStack bookkeeping of this env (since_update_kernel), same launch
    const int p = since_prev ? since_prev[e] : 0;
    since_next[e] = (flags[e] & 2) ? 0 : (uint8_t)(p + 1 > 3 ? 3 : p + 1);
  }
  const int* hdr = (const int*)blob;
  const int* xstart = (const int*)(blob + hdr[3]);
  const int* ystart = (const int*)(blob + hdr[4]);
  const Tap* xt = (const Tap*)(blob + hdr[5]);
  const Tap* yt = (const Tap*)(blob + hdr[6]);
  const uint32_t* pal_g = (const uint32_t*)(blob + hdr[7]) - 128;
  if (threadIdx.x < 128) pal[threadIdx.x] = pal_g[threadIdx.x];
  // the tap tables are read ~30 times per output pixel in dependent chains: stage them in LDS
  // (global / L2 latency per tap made this kernel latency-bound at ~0.5 TB/s)
  for (int i = threadIdx.x; i <= dim; i += blockDim.x) { s_xstart[i] = xstart[i]; s_ystart[i] = ystart[i]; }
  {
    const int nx = xstart[dim], ny = ystart[dim];
    for (int i = threadIdx.x; i < nx; i += blockDim.x) s_xt[i] = xt[i];
    for (int i = threadIdx.x; i < ny; i += blockDim.x) s_yt[i] = yt[i];
  }
  __syncthreads();
  const bool single = (frames1 == nullptr) || (flags && (flags[e] & 1));
  const uint8_t* f0 = frames0 + (size_t)e * in_stride;
  const uint8_t* f1 = single ? f0 : frames1 + (size_t)e * in_stride;
  if (fmt == 1) {
    // gray of ONE colour byte (both frames agree: the static part of every picture) from a 128-entry
    // table built here from the palette with the same integer formula; 16 pixels per lane and step,
    // and a wave whose 1024 pixels all agree never runs the two-colour path (max per channel)
    uint8_t* g1 = (uint8_t*)s_ystart + 4 * (dim + 1);      // [128] bytes behind the tap tables (see lds size)
    if (threadIdx.x < 128) {
      const uint32_t c = pal[threadIdx.x];
      g1[threadIdx.x] = (uint8_t)((((c >> 16) & 255) * 4899u + ((c >> 8) & 255) * 9617u + (c & 255) * 1868u + 8192u) >> 14);
    }
    __syncthreads();
    const uint4* a4 = (const uint4*)f0;
    const uint4* b4 = (const uint4*)f1;
    // (Round 4 measured issuing these loads up front — four passes per thread and frame in registers, in flight during
    // the header / table staging above — on the idea that the kernel is one chain of HBM round trips: A/B on one box
    // 26.4 -> 29.4 us alone and 664 -> 669 us for env step + frame_post; the extra registers (34 -> 63) cost more
    // than the overlap buys, the four resident workgroups per CU already hide each other's loads.  Not kept.)
    for (int i = threadIdx.x; i < kFrameBytes / 16; i += blockDim.x) {
      const uint4 a = a4[i], b = b4[i];
      const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
      uint32_t ow[4];
      const bool same = (((a.x ^ b.x) | (a.y ^ b.y) | (a.z ^ b.z) | (a.w ^ b.w)) & 0xfefefefeu) == 0u;
      if (__ballot(!same) == 0ull) {
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
          for (int j = 0; j < 4; ++j)
            o |= gray_of_colors((aw[q] >> (8 * j)) & 255, (bw[q] >> (8 * j)) & 255, pal) << (8 * j);
          ow[q] = o;
        }
      }
      ((uint4*)gray)[i] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
    }
  } else {
    for (int i = threadIdx.x; i < kFrameBytes; i += blockDim.x) {
      uint32_t r = f0[3 * i], g = f0[3 * i + 1], b = f0[3 * i + 2];
      if (!single) {
        const uint32_t r1 = f1[3 * i], g1 = f1[3 * i + 1], b1 = f1[3 * i + 2];
        r = r1 > r ? r1 : r; g = g1 > g ? g1 : g; b = b1 > b ? b1 : b;
      }
      gray[i] = (uint8_t)((r * 4899u + g * 9617u + b * 1868u + 8192u) >> 14);
    }
  }
  __syncthreads();
  uint8_t* o = out + (size_t)e * out_stride;
  // a thread keeps ONE output column dx (its x taps live in registers) and walks the rows in steps of
  // blockDim / dim; float operation order as in the oracle: buf = sum_k S[x_k] * alpha_k in tap order,
  // sum = beta_0 * buf_0, then += beta_j * buf_j
  const int rows_per_pass = blockDim.x / dim;
  if ((int)threadIdx.x < rows_per_pass * dim && rows_per_pass > 0) {
    const int dy0 = threadIdx.x / dim, dx = threadIdx.x - dy0 * dim;
    const int x0 = s_xstart[dx], nxt = s_xstart[dx + 1] - x0;
    constexpr int kMaxXT = 8;
    int xsi[kMaxXT];
    float xal[kMaxXT];
#pragma unroll
    for (int k = 0; k < kMaxXT; ++k) {
      const Tap tx = s_xt[x0 + (k < nxt ? k : 0)];
      xsi[k] = tx.si;
      xal[k] = tx.alpha;
    }
    for (int dy = dy0; dy < dim; dy += rows_per_pass) {
      const int j0 = s_ystart[dy], j1 = s_ystart[dy + 1];
      float sum = 0.f;
      for (int j = j0; j < j1; ++j) {
        const Tap ty = s_yt[j];
        const uint8_t* S = gray + ty.si * kW;
        float buf = 0.f;
        if (nxt <= kMaxXT) {
#pragma unroll
          for (int k = 0; k < kMaxXT; ++k)
            if (k < nxt) buf = __fadd_rn(buf, __fmul_rn((float)S[xsi[k]], xal[k]));
        } else {  // dim < 20: more than 8 source columns per output column
          for (int k = x0; k < x0 + nxt; ++k) {
            const Tap tx = s_xt[k];
            buf = __fadd_rn(buf, __fmul_rn((float)S[tx.si], tx.alpha));
          }
        }
        const float tmp = __fmul_rn(ty.alpha, buf);
        sum = (j == j0) ? tmp : __fadd_rn(sum, tmp);
      }
      // cv::saturate_cast<uchar>(float): cvRound (round half to even) then clamp
      int r = (int)__builtin_rintf(sum);
      r = r < 0 ? 0 : (r > 255 ? 255 : r);
      o[dy * dim + dx] = (uint8_t)r;
    }
  } else if (rows_per_pass == 0) {  // dim > blockDim: generic walk
    for (int p = threadIdx.x; p < dim * dim; p += blockDim.x) {
      const int dy = p / dim, dx = p - dy * dim;
      const int x0 = s_xstart[dx], x1 = s_xstart[dx + 1];
      const int j0 = s_ystart[dy], j1 = s_ystart[dy + 1];
      float sum = 0.f;
      for (int j = j0; j < j1; ++j) {
        const Tap ty = s_yt[j];
        const uint8_t* S = gray + ty.si * kW;
        float buf = 0.f;
        for (int k = x0; k < x1; ++k) {
          const Tap tx = s_xt[k];
          buf = __fadd_rn(buf, __fmul_rn((float)S[tx.si], tx.alpha));
        }
        const float tmp = __fmul_rn(ty.alpha, buf);
        sum = (j == j0) ? tmp : __fadd_rn(sum, tmp);
      }
      int r = (int)__builtin_rintf(sum);
      r = r < 0 ? 0 : (r > 255 ? 255 : r);
      o[p] = (uint8_t)r;
    }
  }
}

// FrameStack (atari_wrappers.py:270-306) without storing stacks: the rollout ring keeps ONE
// dim*dim frame per (slot, env); a stacked obs is gathered as channel j = ring[slot -
// min(3-j, since)] where `since` = steps since the env's last reset (0 => 4 copies, :290-294).
__global__ __launch_bounds__(256) void stack_gather_kernel(
    const uint8_t* __restrict__ ring, const uint8_t* __restrict__ since, int E, int fsz,
    const int* __restrict__ slots, const int* __restrict__ envs, int64_t n,
    uint8_t* __restrict__ out, int num_slots /* > 0: the ring is circular, slot - back wraps */,
    const int* __restrict__ link /* [S,E] or null: slot of the env's PREVIOUS observation (elastic
                                    launches leave gaps); null = the slot before */) {
  // one workgroup per (sample, channel); 16-byte copies
  const int64_t s = blockIdx.x >> 2;
  const int j = blockIdx.x & 3;
  if (s >= n) return;
  const int slot = slots[s];
  const int env = envs[s];
  int back = 3 - j;
  const int sr = since[(size_t)slot * E + env];
  back = back < sr ? back : sr;
  int from = slot;
  if (link) {
    for (int k = 0; k < back; ++k) from = link[(size_t)from * E + env];
  } else {
    from = slot - back;
    if (from < 0) from += num_slots;
  }
  const uint8_t* src = ring + ((size_t)from * E + env) * fsz;
  uint8_t* dst = out + ((size_t)s * 4 + j) * fsz;
  if ((fsz & 15) == 0) {
    for (int i = threadIdx.x; i < fsz / 16; i += blockDim.x) ((uint4*)dst)[i] = ((const uint4*)src)[i];
  } else {
    for (int i = threadIdx.x; i < fsz / 4; i += blockDim.x) ((uint32_t*)dst)[i] = ((const uint32_t*)src)[i];
  }
}

// since[slot+1][e] = reset ? 0 : min(since[slot][e] + 1, 3)   (flags bit1 = reset this step)
__global__ void since_update_kernel(const uint8_t* __restrict__ flags,
                                    const uint8_t* __restrict__ since_prev,
                                    uint8_t* __restrict__ since_next, int E) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int p = since_prev ? since_prev[e] : 0;
  since_next[e] = (flags[e] & 2) ? 0 : (uint8_t)(p + 1 > 3 ? 3 : p + 1);
}

// MonitorEnv bookkeeping reduced on the device: acc[0] += #episodes closed this step,
// acc[1] += their unclipped returns, acc[2] += their lengths (few lanes ever take the atomics)
__global__ void episode_stats_kernel(const float* __restrict__ ep_returns, const int* __restrict__ ep_lengths,
                                     int E, double* __restrict__ acc) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int len = ep_lengths[e];
  if (len > 0) {
    atomicAdd(acc + 0, 1.0);
    atomicAdd(acc + 1, (double)ep_returns[e]);
    atomicAdd(acc + 2, (double)len);
  }
}

}  // namespace atari
}  // namespace parlhip

using namespace parlhip;
using namespace parlhip::atari;

// ---- host-side frame_post tables (same construction as oracle/frame_oracle.c; OpenCV
//      computeResizeAreaTab restated) ----
namespace {
int area_tab(int ssize, int dsize, int* start, Tap* tab) {
  const double inv = (double)dsize / (double)ssize;
  const double scale = 1.0 / inv;
  int k = 0;
  for (int dx = 0; dx < dsize; ++dx) {
    start[dx] = k;
    const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
    const double cell = scale < (ssize - fsx1) ? scale : (ssize - fsx1);
    int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
    if (sx2 > ssize - 1) sx2 = ssize - 1;
    if (sx1 > sx2) sx1 = sx2;
    if (sx1 - fsx1 > 1e-3) { if (tab) { tab[k].si = sx1 - 1; tab[k].alpha = (float)((sx1 - fsx1) / cell); } k++; }
    for (int sx = sx1; sx < sx2; ++sx) { if (tab) { tab[k].si = sx; tab[k].alpha = (float)(1.0 / cell); } k++; }
    if (fsx2 - sx2 > 1e-3) {
      double r = fsx2 - sx2;
      if (r > 1.0) r = 1.0;
      if (r > cell) r = cell;
      if (tab) { tab[k].si = sx2; tab[k].alpha = (float)(r / cell); }
      k++;
    }
  }
  start[dsize] = k;
  return k;
}
// Stella 2.x NTSC palette (colour byte >> 1 -> 0xRRGGBB)
const uint32_t k_ntsc[128] = {
    0x000000, 0x4a4a4a, 0x6f6f6f, 0x8e8e8e, 0xaaaaaa, 0xc0c0c0, 0xd6d6d6, 0xececec, 0x484800, 0x69690f, 0x86861d,
    0xa2a22a, 0xbbbb35, 0xd2d240, 0xe8e84a, 0xfcfc54, 0x7c2c00, 0x904811, 0xa26221, 0xb47a30, 0xc3903d, 0xd2a44a,
    0xdfb755, 0xecc860, 0x901c00, 0xa33915, 0xb55328, 0xc66c3a, 0xd5824a, 0xe39759, 0xf0aa67, 0xfcbc74, 0x940000,
    0xa71a1a, 0xb83232, 0xc84848, 0xd65c5c, 0xe46f6f, 0xf08080, 0xfc9090, 0x840064, 0x97197a, 0xa8308f, 0xb846a2,
    0xc659b3, 0xd46cc3, 0xe07cd2, 0xec8ce0, 0x500084, 0x68199a, 0x7d30ad, 0x9246c0, 0xa459d0, 0xb56ce0, 0xc57cee,
    0xd48cfc, 0x140090, 0x331aa3, 0x4e32b5, 0x6848c6, 0x7f5cd5, 0x956fe3, 0xa980f0, 0xbc90fc, 0x000094, 0x181aa7,
    0x2d32b8, 0x4248c8, 0x545cd6, 0x656fe4, 0x7580f0, 0x8490fc, 0x001c88, 0x183b9d, 0x2d57b0, 0x4272c2, 0x548ad2,
    0x65a0e1, 0x75b5ef, 0x84c8fc, 0x003064, 0x185080, 0x2d6d98, 0x4288b0, 0x54a0c5, 0x65b7d9, 0x75cceb, 0x84e0fc,
    0x004030, 0x18624e, 0x2d8169, 0x429e82, 0x54b899, 0x65d1ae, 0x75e7c2, 0x84fcd4, 0x004400, 0x1a661a, 0x328432,
    0x48a048, 0x5cba5c, 0x6fd26f, 0x80e880, 0x90fc90, 0x143c00, 0x355f18, 0x527e2d, 0x6e9c42, 0x87b754, 0x9ed065,
    0xb4e775, 0xc8fc84, 0x303800, 0x505916, 0x6d762b, 0x88923e, 0xa0ab4f, 0xb7c25f, 0xccd86e, 0xe0ec7c, 0x482c00,
    0x694d14, 0x866a26, 0xa28638, 0xbb9f47, 0xd2b656, 0xe8cc63, 0xfce070};
}  // namespace

PARLHIP_EXPORT size_t parlhip_frame_post_tables_bytes(int dim) {
  if (dim < 1 || dim > 210) return 0;
  std::vector<int> tmp(dim + 1);
  const int nx = area_tab(kW, dim, tmp.data(), nullptr);
  const int ny = area_tab(kH, dim, tmp.data(), nullptr);
  return 8 * 4 + (size_t)tail_lane_taps_bytes(dim) + 2 * (size_t)(dim + 1) * 4 + (size_t)(nx + ny) * sizeof(Tap) + 128 * 4;
}

PARLHIP_EXPORT int parlhip_frame_post_tables_init(void* host_blob, int dim) {
  if (!host_blob || dim < 1 || dim > 210) return PARLHIP_EINVAL;
  int* hdr = (int*)host_blob;
  Tap* lane_taps = (Tap*)(hdr + 8);
  int* xstart = (int*)((char*)(hdr + 8) + tail_lane_taps_bytes(dim));
  int* ystart = xstart + dim + 1;
  Tap* xt = (Tap*)(ystart + dim + 1);
  const int nx = area_tab(kW, dim, xstart, xt);
  Tap* yt = xt + nx;
  const int ny = area_tab(kH, dim, ystart, yt);
  uint32_t* pal = (uint32_t*)(yt + ny);
  memcpy(pal, k_ntsc, sizeof(k_ntsc));
  hdr[0] = dim; hdr[1] = nx; hdr[2] = ny;
  hdr[3] = (int)((char*)xstart - (char*)host_blob);
  hdr[4] = (int)((char*)ystart - (char*)host_blob);
  hdr[5] = (int)((char*)xt - (char*)host_blob);
  hdr[6] = (int)((char*)yt - (char*)host_blob);
  hdr[7] = (int)((char*)(pal + 128) - (char*)host_blob);
  if (tail_lane_taps_bytes(dim)) {   // the observation tail's lane-ordered copy (frame_defs.hpp)
    const int NX = tail_lane_taps_nx(dim), NC = tail_lane_taps_nc(dim), M = dim / 42, NY = dim == 42 ? 5 : 3;
    for (int c = 0; c < NC; ++c)
      for (int k = 0; k < NX; ++k)
        for (int lane = 0; lane < 64; ++lane) {
          const int dx = lane + 64 * c;
          Tap t = {0, 0.f};
          if (dx < dim) {
            const int x0 = xstart[dx], n = xstart[dx + 1] - x0;
            if (n > NX) return PARLHIP_EINVAL;
            t.si = xt[x0 + (k < n ? k : 0)].si;
            t.alpha = k < n ? xt[x0 + k].alpha : 0.f;
          }
          lane_taps[(c * NX + k) * 64 + lane] = t;
        }
    Tap* ylane = lane_taps + NC * NX * 64;
    for (int j = 0; j < 8; ++j) ylane[j] = j < M * NY ? yt[j] : Tap{0, 0.f};
  }
  return PARLHIP_OK;
}

PARLHIP_EXPORT int parlhip_frame_post_u8(const uint8_t* frames0, const uint8_t* frames1,
                                         int64_t in_stride, int fmt, const uint8_t* flags,
                                         uint8_t* out, int64_t out_stride, int E, int dim,
                                         const void* tables_dev, parlhip_stream_t stream) {
  if (E < 0 || dim < 1 || dim > 210 || (fmt != 0 && fmt != 1)) return PARLHIP_EINVAL;
  if (E == 0) return PARLHIP_OK;
  if (!frames0 || !out || !tables_dev) return PARLHIP_EINVAL;
  if (fmt == 1 && ((reinterpret_cast<uintptr_t>(frames0) | (uintptr_t)in_stride |
                    (frames1 ? reinterpret_cast<uintptr_t>(frames1) : 0)) & 15))
    return PARLHIP_EINVAL;
  const size_t lds = frame_post_lds_bytes(dim);
  if (lds > 48 * 1024) {  // dim > ~84: beyond the default dynamic-LDS limit, raise it once
    static int raised_for = 0;
    if (raised_for < (int)lds) {
      int rc = check(hipFuncSetAttribute((const void*)frame_post_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds));
      if (rc) return rc;
      raised_for = (int)lds;
    }
  }
  frame_post_kernel<<<E, 512, lds, (hipStream_t)stream>>>(frames0, frames1, in_stride, fmt, flags, out,
                                                          out_stride, dim, (const uint8_t*)tables_dev, nullptr, nullptr,
                                                          nullptr, nullptr, nullptr);
  return check_launch();
}

PARLHIP_EXPORT int parlhip_frame_post_since_u8(const uint8_t* frames0, const uint8_t* frames1, int64_t in_stride,
                                               int fmt, const uint8_t* flags, uint8_t* out, int64_t out_stride, int E,
                                               int dim, const void* tables_dev, const uint8_t* since_prev,
                                               uint8_t* since_next, parlhip_stream_t stream) {
  if (E < 0 || dim < 1 || dim > 84 || (fmt != 0 && fmt != 1)) return PARLHIP_EINVAL;
  if (E == 0) return PARLHIP_OK;
  if (!frames0 || !out || !tables_dev || !flags || !since_next) return PARLHIP_EINVAL;
  if (fmt == 1 && ((reinterpret_cast<uintptr_t>(frames0) | (uintptr_t)in_stride |
                    (frames1 ? reinterpret_cast<uintptr_t>(frames1) : 0)) & 15))
    return PARLHIP_EINVAL;
  frame_post_kernel<<<E, 512, frame_post_lds_bytes(dim), (hipStream_t)stream>>>(
      frames0, frames1, in_stride, fmt, flags, out, out_stride, dim, (const uint8_t*)tables_dev, since_prev, since_next,
      nullptr, nullptr, nullptr);
  return check_launch();
}

PARLHIP_EXPORT int parlhip_frame_post_step_u8(const uint8_t* frames0, const uint8_t* frames1, int64_t in_stride,
                                              int fmt, const uint8_t* flags, uint8_t* out, int64_t out_stride, int E,
                                              int dim, const void* tables_dev, const uint8_t* since_prev,
                                              uint8_t* since_next, const float* ep_returns, const int32_t* ep_lengths,
                                              double* ep_acc3, parlhip_stream_t stream) {
  if (E < 0 || dim < 1 || dim > 84 || (fmt != 0 && fmt != 1)) return PARLHIP_EINVAL;
  if (E == 0) return PARLHIP_OK;
  if (!frames0 || !out || !tables_dev || !flags || !since_next || !ep_returns || !ep_lengths || !ep_acc3)
    return PARLHIP_EINVAL;
  if (fmt == 1 && ((reinterpret_cast<uintptr_t>(frames0) | (uintptr_t)in_stride |
                    (frames1 ? reinterpret_cast<uintptr_t>(frames1) : 0)) & 15))
    return PARLHIP_EINVAL;
  frame_post_kernel<<<E, 512, frame_post_lds_bytes(dim), (hipStream_t)stream>>>(
      frames0, frames1, in_stride, fmt, flags, out, out_stride, dim, (const uint8_t*)tables_dev, since_prev, since_next,
      ep_returns, ep_lengths, ep_acc3);
  return check_launch();
}

PARLHIP_EXPORT int parlhip_stack_since_update_u8(const uint8_t* obs_flags, const uint8_t* since_prev,
                                                 uint8_t* since_next, int E, parlhip_stream_t stream) {
  if (E < 0) return PARLHIP_EINVAL;
  if (E == 0) return PARLHIP_OK;
  if (!obs_flags || !since_next) return PARLHIP_EINVAL;
  since_update_kernel<<<ceil_div(E, 256), 256, 0, (hipStream_t)stream>>>(obs_flags, since_prev, since_next, E);
  return check_launch();
}

PARLHIP_EXPORT int parlhip_stack_gather_u8(const uint8_t* ring, const uint8_t* since, int E, int frame_bytes,
                                           const int32_t* slots, const int32_t* envs, int64_t n,
                                           uint8_t* out, parlhip_stream_t stream) {
  if (E < 1 || frame_bytes < 4 || (frame_bytes & 3) || n < 0) return PARLHIP_EINVAL;
  if (n == 0) return PARLHIP_OK;
  if (!ring || !since || !out || !slots || !envs) return PARLHIP_EINVAL;
  if (n * 4 > 0x7fffffffLL) return PARLHIP_ENOSUP;
  stack_gather_kernel<<<(unsigned)(n * 4), 256, 0, (hipStream_t)stream>>>(ring, since, E, frame_bytes, slots,
                                                                          envs, n, out, 0, nullptr);
  return check_launch();
}

PARLHIP_EXPORT int parlhip_stack_gather_ring_u8(const uint8_t* ring, const uint8_t* since, const int32_t* link,
                                                int num_slots, int E, int frame_bytes, const int32_t* slots,
                                                const int32_t* envs, int64_t n, uint8_t* out,
                                                parlhip_stream_t stream) {
  if (E <= 0 || frame_bytes <= 0 || (frame_bytes & 3) || n < 0 || num_slots < 4) return PARLHIP_EINVAL;
  if (n == 0) return PARLHIP_OK;
  if (!ring || !since || !out || !slots || !envs) return PARLHIP_EINVAL;
  if (n * 4 > 0x7fffffffLL) return PARLHIP_ENOSUP;
  stack_gather_kernel<<<(unsigned)(n * 4), 256, 0, (hipStream_t)stream>>>(ring, since, E, frame_bytes, slots,
                                                                          envs, n, out, num_slots, link);
  return check_launch();
}

PARLHIP_EXPORT int parlhip_episode_stats_accum_f64(const float* ep_returns, const int32_t* ep_lengths, int E,
                                                   double* acc3, parlhip_stream_t stream) {
  if (E < 0) return PARLHIP_EINVAL;
  if (E == 0) return PARLHIP_OK;
  if (!ep_returns || !ep_lengths || !acc3) return PARLHIP_EINVAL;
  episode_stats_kernel<<<ceil_div(E, 256), 256, 0, (hipStream_t)stream>>>(ep_returns, ep_lengths, E, acc3);
  return check_launch();
}
