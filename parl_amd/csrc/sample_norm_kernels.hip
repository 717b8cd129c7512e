// sample_norm_kernels.hip — categorical action sampling and PPO advantage normalisation.
//
// Reference arithmetic (paths relative to the PARL tree):
//   examples/IMPALA/atari_agent.py:38-40   np.random.choice(len(prob), 1, p=prob) per env
//   parl/algorithms/paddle/impala/impala.py:217-227   IMPALA.sample: softmax(policy(obs))
//   parl/algorithms/torch/ppo.py:115-117 / paddle/ppo.py:124-127   (adv-mean)/(std+1e-8)
#include "common.hpp"
#include "philox.hpp"
#include "policy_head.hpp"
#include <math.h>

namespace parlhip {

// searchsorted(cumsum_f64(p)/cdf[-1], u, side='right')
__device__ __forceinline__ int64_t choice_row(const float* __restrict__ p, int A, double u) {
  double last = 0.0;
  for (int k = 0; k < A; ++k) last += (double)p[k];
  double s = 0.0;
  int64_t a = A;
  for (int k = 0; k < A; ++k) {
    s += (double)p[k];
    if (s / last > u) { a = k; break; }
  }
  return a;
}

__global__ __launch_bounds__(256) void categorical_sample_kernel(
    const float* __restrict__ probs, const double* __restrict__ uniforms,
    int64_t* __restrict__ actions, int B, int A) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  actions[b] = choice_row(probs + b * A, A, uniforms[b]);
}

__global__ __launch_bounds__(256) void policy_sample_kernel(
    const float* __restrict__ x, int is_logits, int64_t* __restrict__ actions,
    float* __restrict__ probs_out, double* __restrict__ uniforms_out, int B, int A,
    uint64_t seed, uint64_t offset, uint64_t row0, const uint64_t* __restrict__ offset_base) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  if (offset_base) offset += *offset_base;   // the rollout's first step lives in device memory (hipGraph replays)
  const float* row = x + b * A;
  const double u = philox_uniform53(seed, offset, row0 + (uint64_t)b);
  if (uniforms_out) uniforms_out[b] = u;
  if (!is_logits) {
    actions[b] = choice_row(row, A, u);
    if (probs_out)
      for (int k = 0; k < A; ++k) probs_out[b * A + k] = row[k];
    return;
  }
  // softmax in float32 (F.softmax), max-subtracted
  float m = row[0];
  for (int k = 1; k < A; ++k) m = fmaxf(m, row[k]);
  float s = 0.f;
  for (int k = 0; k < A; ++k) s += expf(row[k] - m);
  // pass 2: cdf in float64 of the float32 probabilities e_k / s
  double last = 0.0;
  for (int k = 0; k < A; ++k) last += (double)(expf(row[k] - m) / s);
  double c = 0.0;
  int64_t a = A;
  for (int k = 0; k < A; ++k) {
    const float pk = expf(row[k] - m) / s;
    if (probs_out) probs_out[b * A + k] = pk;
    c += (double)pk;
    if (a == A && c / last > u) a = k;
  }
  actions[b] = a;
}

// The actors' policy head + sampling in ONE launch (examples/IMPALA/atari_model.py:44-57 policy_fc, then
// IMPALA.sample + AtariAgent.sample, atari_agent.py:35-42): one wavefront per env row — lane l holds
// h[row][4l..4l+3], a logit is four FMAs per lane and a wave reduction, lane 0 adds the bias, writes the
// [A] logits row into the rollout slab and draws the action exactly as policy_sample_kernel does
// (float32 softmax, float64 inverse CDF, Philox uniform of (offset, row0 + row)).  256 hidden units.
template <int A_MAX>
__global__ __launch_bounds__(256) void policy_head_sample_kernel(
    const float* __restrict__ h, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ logits_out, int64_t* __restrict__ actions, int B, int A, uint64_t seed, uint64_t offset,
    uint64_t row0, const uint64_t* __restrict__ offset_base) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;  // whole wave
  if (offset_base) offset += *offset_base;  // the rollout's first step lives in device memory (hipGraph replays)
  float row[A_MAX];
  policy_head_row<A_MAX>(h + b * 256, w, bias, A, lane, row);
  if (lane != 0) return;
  const double u = philox_uniform53(seed, offset, row0 + (uint64_t)b);
#pragma unroll
  for (int k = 0; k < A_MAX; ++k) if (k < A) logits_out[b * A + k] = row[k];
  const int64_t a = policy_draw<A_MAX>(row, A, u);
  actions[b] = a;
}

// ----------------------------------------------------------------------------------------
// Advantage normalisation: two kernels, deterministic (no atomics).
//   pass 1: per-block partial (sum, sum of squares about a pivot) in float64 -> workspace
//   pass 2: every block re-reduces the partials (<= 1024 pairs), then normalises its slice.
// Using a pivot (the first element) keeps sum-of-squares well conditioned.
// ----------------------------------------------------------------------------------------
constexpr int kNormBlocksMax = 1024;

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
  return x;
}

__global__ __launch_bounds__(256) void adv_partial_kernel(const float* __restrict__ adv,
                                                          const int64_t* __restrict__ idx,
                                                          int64_t n, double* __restrict__ ws) {
  __shared__ double sh[2][4];
  const float pivot = adv[idx ? idx[0] : 0];
  double s = 0.0, q = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const double d = (double)(adv[idx ? idx[i] : i] - pivot);
    s += d;
    q += d * d;
  }
  s = wave_sum(s);
  q = wave_sum(q);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sh[0][w] = s; sh[1][w] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    ws[2 * blockIdx.x] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
    ws[2 * blockIdx.x + 1] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
  }
}

__global__ __launch_bounds__(256) void adv_apply_kernel(
    const float* __restrict__ adv, const int64_t* __restrict__ idx, float* __restrict__ out,
    int64_t n, float eps, const double* __restrict__ ws, int nparts,
    float* __restrict__ mean_std_out) {
  __shared__ double sh[2][4];
  __shared__ float sh_mean, sh_std;
  double s = 0.0, q = 0.0;
  for (int p = threadIdx.x; p < nparts; p += blockDim.x) { s += ws[2 * p]; q += ws[2 * p + 1]; }
  s = wave_sum(s);
  q = wave_sum(q);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sh[0][w] = s; sh[1][w] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double S = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
    const double Q = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
    const double pivot = (double)adv[idx ? idx[0] : 0];
    const double dm = S / (double)n;
    const double var = n > 1 ? (Q - S * dm) / (double)(n - 1) : __builtin_nan("");
    sh_mean = (float)(pivot + dm);
    sh_std = (float)sqrt(var < 0.0 ? 0.0 : var);
    if (blockIdx.x == 0 && mean_std_out) { mean_std_out[0] = sh_mean; mean_std_out[1] = sh_std; }
  }
  __syncthreads();
  const float mean = sh_mean, denom = sh_std + eps;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (adv[idx ? idx[i] : i] - mean) / denom;
}

static inline int norm_blocks(int64_t n) {
  int64_t b = (n + 256 * 8 - 1) / (256 * 8);  // >= 8 elements per thread
  if (b < 1) b = 1;
  if (b > kNormBlocksMax) b = kNormBlocksMax;
  return (int)b;
}

}  // namespace parlhip

using namespace parlhip;

PARLHIP_EXPORT int parlhip_categorical_sample_f32(const float* probs, const double* uniforms,
                                              int64_t* actions, int B, int A,
                                              parlhip_stream_t stream) {
  if (B < 0 || A < 1) return PARLHIP_EINVAL;
  if (B == 0) return PARLHIP_OK;
  if (!probs || !uniforms || !actions) return PARLHIP_EINVAL;
  const int block = B >= 256 * 64 ? 256 : 64;
  categorical_sample_kernel<<<ceil_div(B, block), block, 0, (hipStream_t)stream>>>(
      probs, uniforms, actions, B, A);
  return check_launch();
}

static int launch_policy_head_sample(const float* hidden, const float* w_policy, const float* b_policy,
                                     float* logits_out, int64_t* actions, int B, int hidden_units, int A,
                                     uint64_t seed, uint64_t offset, uint64_t row0, const uint64_t* offset_base,
                                     parlhip_stream_t stream) {
  if (B < 0 || A < 1) return PARLHIP_EINVAL;
  if (hidden_units != 256 || A > 18) return PARLHIP_ENOSUP;
  if (B == 0) return PARLHIP_OK;
  if (!hidden || !w_policy || !b_policy || !logits_out || !actions) return PARLHIP_EINVAL;
  if ((reinterpret_cast<uintptr_t>(hidden) | reinterpret_cast<uintptr_t>(w_policy)) & 15) return PARLHIP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (A <= 6)
    policy_head_sample_kernel<6><<<ceil_div(B, 4), 256, 0, s>>>(hidden, w_policy, b_policy, logits_out, actions, B, A,
                                                                 seed, offset, row0, offset_base);
  else
    policy_head_sample_kernel<18><<<ceil_div(B, 4), 256, 0, s>>>(hidden, w_policy, b_policy, logits_out, actions, B, A,
                                                                  seed, offset, row0, offset_base);
  return check_launch();
}

PARLHIP_EXPORT int parlhip_policy_head_sample_f32(const float* hidden, const float* w_policy, const float* b_policy,
                                                  float* logits_out, int64_t* actions, int B, int hidden_units, int A,
                                                  uint64_t seed, uint64_t offset, uint64_t row0,
                                                  parlhip_stream_t stream) {
  return launch_policy_head_sample(hidden, w_policy, b_policy, logits_out, actions, B, hidden_units, A, seed, offset,
                                   row0, nullptr, stream);
}

PARLHIP_EXPORT int parlhip_policy_head_sample_at_f32(const float* hidden, const float* w_policy, const float* b_policy,
                                                     float* logits_out, int64_t* actions, int B, int hidden_units,
                                                     int A, uint64_t seed, const uint64_t* offset_base,
                                                     uint64_t offset, uint64_t row0, parlhip_stream_t stream) {
  if (!offset_base) return PARLHIP_EINVAL;
  return launch_policy_head_sample(hidden, w_policy, b_policy, logits_out, actions, B, hidden_units, A, seed, offset,
                                   row0, offset_base, stream);
}

static int launch_policy_sample(const float* x, int is_logits, int64_t* actions, float* probs_out, double* uniforms_out,
                                int B, int A, uint64_t seed, uint64_t offset, uint64_t row0, const uint64_t* offset_base,
                                parlhip_stream_t stream) {
  if (B < 0 || A < 1) return PARLHIP_EINVAL;
  if (B == 0) return PARLHIP_OK;
  if (!x || !actions) return PARLHIP_EINVAL;
  const int block = B >= 256 * 64 ? 256 : 64;
  policy_sample_kernel<<<ceil_div(B, block), block, 0, (hipStream_t)stream>>>(
      x, is_logits, actions, probs_out, uniforms_out, B, A, seed, offset, row0, offset_base);
  return check_launch();
}

PARLHIP_EXPORT int parlhip_policy_sample_f32(const float* x, int is_logits, int64_t* actions,
                                         float* probs_out, double* uniforms_out, int B, int A,
                                         uint64_t seed, uint64_t offset, uint64_t row0,
                                         parlhip_stream_t stream) {
  return launch_policy_sample(x, is_logits, actions, probs_out, uniforms_out, B, A, seed, offset, row0, nullptr, stream);
}

PARLHIP_EXPORT int parlhip_policy_sample_at_f32(const float* x, int is_logits, int64_t* actions, float* probs_out,
                                                double* uniforms_out, int B, int A, uint64_t seed,
                                                const uint64_t* offset_base, uint64_t offset, uint64_t row0,
                                                parlhip_stream_t stream) {
  if (!offset_base) return PARLHIP_EINVAL;
  return launch_policy_sample(x, is_logits, actions, probs_out, uniforms_out, B, A, seed, offset, row0, offset_base,
                              stream);
}

PARLHIP_EXPORT size_t parlhip_adv_normalize_workspace_bytes(int64_t n) {
  return (size_t)norm_blocks(n < 0 ? 0 : n) * 2 * sizeof(double);
}

PARLHIP_EXPORT int parlhip_adv_normalize_f32(const float* adv, const int64_t* idx, float* out,
                                         int64_t n, float eps, void* workspace,
                                         size_t workspace_bytes, float* mean_std_out,
                                         parlhip_stream_t stream) {
  if (n < 0) return PARLHIP_EINVAL;
  if (n == 0) return PARLHIP_OK;
  if (!adv || !out || !workspace) return PARLHIP_EINVAL;
  const int nb = norm_blocks(n);
  if (workspace_bytes < (size_t)nb * 2 * sizeof(double)) return PARLHIP_ENOMEM;
  if (reinterpret_cast<uintptr_t>(workspace) & 7) return PARLHIP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  adv_partial_kernel<<<nb, 256, 0, s>>>(adv, idx, n, (double*)workspace);
  int rc = check_launch();
  if (rc) return rc;
  adv_apply_kernel<<<nb, 256, 0, s>>>(adv, idx, out, n, eps, (const double*)workspace, nb,
                                      mean_std_out);
  return check_launch();
}
