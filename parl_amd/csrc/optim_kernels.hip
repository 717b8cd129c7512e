// optim_kernels.hip — the learner's parameter update: global-norm gradient clipping + Adam, two launches.
//
// Reference: parl/algorithms/paddle/impala.py:113-117 (`Adam(learning_rate, grad_clip=ClipGradByGlobalNorm(40))`),
// parl/algorithms/torch/a2c.py:76-78 (`clip_grad_norm_(parameters, 40)` then `optimizer.step()`).  The host mirror
// expresses the pair as torch.nn.utils.clip_grad_norm_ + torch.optim.Adam.step, which on a 1 M-parameter model is
// ~12 small launches and one slow one: torch's fused multi-tensor Adam hands each workgroup a 65,536-element chunk
// — 16 workgroups on 256 CUs, 39 us for 16 MB of traffic — and the norm / clamp / scale chain around it is another
// ~45 us of launch-bound kernels.  At the reference's learner batch (1000 rows, one hipGraph replay of ~390 us per
// update) that is a fifth of the update.  Here:
//   launch 1: every workgroup sums g^2 over its 2,048-element chunk -> partial[block] (no atomics), and the first
//             workgroup of every tensor advances that tensor's step counter;
//   launch 2: every workgroup adds the partials in one fixed order (all workgroups get the same bits), forms
//             clip = min(1, max_norm / (norm + 1e-6)), scales its chunk of g in place and applies Adam to it.
// Arithmetic per element is torch's (`_fused_adam` with capturable step / lr tensors): m += (1-b1)(g-m),
// v = b2 v + (1-b2) g^2, p -= (lr / (1-b1^t)) * m / (sqrt(v) / sqrt(1-b2^t) + eps); the global norm is the root of
// the sum of squares of all elements (torch: the norm of the per-tensor norms — the same number up to rounding).
// Deterministic; no host synchronisation; capturable in a hipGraph (pointers and scalars are kernel arguments).
#include "common.hpp"

namespace parlhip {

constexpr int kOptMaxTensors = 16;
constexpr int kOptChunk = 2048;      // elements per workgroup: 256 threads x two float4

struct OptTable {
  float* p[kOptMaxTensors];
  float* g[kOptMaxTensors];
  float* m[kOptMaxTensors];
  float* v[kOptMaxTensors];
  float* step[kOptMaxTensors];
  long long numel[kOptMaxTensors];
  int first_block[kOptMaxTensors + 1];   // prefix sums of ceil(numel / kOptChunk)
  int n;
};

// which tensor a workgroup works on (wave-uniform: scalar compares over <= 16 prefix entries)
__device__ __forceinline__ int opt_tensor_of(const OptTable& t, int block) {
  int k = 0;
#pragma unroll
  for (int i = 1; i < kOptMaxTensors; ++i) k += (i < t.n && block >= t.first_block[i]) ? 1 : 0;
  return k;
}

__device__ __forceinline__ float block_sum_256(float x, float* red) {   // fixed tree: the same bits in every block
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[wave] = x;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void grad_sqsum_kernel(OptTable t, float* __restrict__ partial) {
  __shared__ float red[4];
  const int k = opt_tensor_of(t, blockIdx.x);
  const long long base = (long long)(blockIdx.x - t.first_block[k]) * kOptChunk;
  const long long n = t.numel[k];
  const float* __restrict__ g = t.g[k];
  float s = 0.f;
  const long long i0 = base + 4 * (long long)threadIdx.x;
  if ((((uintptr_t)g) & 15u) == 0 && base + kOptChunk <= n) {
    const float4 a = *reinterpret_cast<const float4*>(g + i0);
    const float4 b = *reinterpret_cast<const float4*>(g + i0 + 1024);
    s = ((a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w)) + ((b.x * b.x + b.y * b.y) + (b.z * b.z + b.w * b.w));
  } else {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long long i = i0 + 1024 * h + j;
        if (i < n) { const float x = g[i]; s += x * x; }
      }
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = s;
    if (base == 0) *t.step[k] += 1.0f;     // torch keeps `step` as a float32 scalar tensor per parameter
  }
}

__global__ __launch_bounds__(256) void clip_adam_kernel(OptTable t, const float* __restrict__ partial, int n_blocks,
                                                        const float* __restrict__ lr_dev, double beta1, double beta2,
                                                        float eps, float max_norm, float* __restrict__ norm_out) {
  __shared__ float red[4];
  // the global norm: all partials in one fixed order
  float s0 = 0.f, s1 = 0.f;
  int i = threadIdx.x;
  for (; i + 256 < n_blocks; i += 512) { s0 += partial[i]; s1 += partial[i + 256]; }
  if (i < n_blocks) s0 += partial[i];
  const float total = block_sum_256(s0 + s1, red);
  const float norm = sqrtf(total);
  float clip = max_norm / (norm + 1e-6f);
  // torch.clamp(clip_coef, max=1.0) keeps a NaN (a non-finite gradient norm): every gradient, and with it every
  // parameter, turns NaN at once, as after torch's clip_grad_norm_ — a partly poisoned model is harder to notice
  clip = clip < 1.0f ? clip : (clip != clip ? clip : 1.0f);
  if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) *norm_out = norm;

  const int k = opt_tensor_of(t, blockIdx.x);
  const long long base = (long long)(blockIdx.x - t.first_block[k]) * kOptChunk;
  const long long n = t.numel[k];
  float* __restrict__ p = t.p[k];
  float* __restrict__ g = t.g[k];
  float* __restrict__ m = t.m[k];
  float* __restrict__ v = t.v[k];
  // the scalars of the step in double, as torch forms them from its Python-float hyper-parameters (1 - 0.999 in
  // float arithmetic is 1.3e-5 off the float nearest to 0.001); the element arithmetic is float
  const double step = (double)*t.step[k];              // already advanced by grad_sqsum_kernel
  const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
  const float step_size = (float)((double)*lr_dev / bc1), bc2_sqrt = (float)sqrt(bc2);
  const float w1 = (float)(1.0 - beta1), w2 = (float)(1.0 - beta2), b2f = (float)beta2;
  auto upd = [&](float& pp, float& gg, float& mm, float& vv) {
    gg *= clip;
    mm = mm + w1 * (gg - mm);
    vv = b2f * vv + w2 * gg * gg;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    pp = pp - step_size * (mm / denom);
  };
  const long long i0 = base + 4 * (long long)threadIdx.x;
  const bool al = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15u) == 0;
  if (al && base + kOptChunk <= n) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const long long o = i0 + 1024 * h;
      float4 P = *reinterpret_cast<float4*>(p + o), G = *reinterpret_cast<float4*>(g + o);
      float4 M = *reinterpret_cast<float4*>(m + o), V = *reinterpret_cast<float4*>(v + o);
      upd(P.x, G.x, M.x, V.x);
      upd(P.y, G.y, M.y, V.y);
      upd(P.z, G.z, M.z, V.z);
      upd(P.w, G.w, M.w, V.w);
      *reinterpret_cast<float4*>(p + o) = P;
      *reinterpret_cast<float4*>(g + o) = G;
      *reinterpret_cast<float4*>(m + o) = M;
      *reinterpret_cast<float4*>(v + o) = V;
    }
  } else {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long long o = i0 + 1024 * h + j;
        if (o < n) upd(p[o], g[o], m[o], v[o]);
      }
  }
}

}  // namespace parlhip

using namespace parlhip;

static int opt_blocks(int n_tensors, const int64_t* numel) {
  long long b = 0;
  for (int i = 0; i < n_tensors; ++i) b += (numel[i] + kOptChunk - 1) / kOptChunk;
  return b > 0x7fffffffLL ? -1 : (int)b;
}

PARLHIP_EXPORT size_t parlhip_clip_adam_workspace_bytes(int n_tensors, const int64_t* numel) {
  if (n_tensors <= 0 || n_tensors > kOptMaxTensors || !numel) return 0;
  const int b = opt_blocks(n_tensors, numel);
  return b <= 0 ? 0 : (size_t)b * sizeof(float);
}

PARLHIP_EXPORT int parlhip_clip_adam_f32(int n_tensors, float* const* params, float* const* grads,
                                         float* const* exp_avg, float* const* exp_avg_sq, float* const* steps,
                                         const int64_t* numel, const float* lr, double beta1, double beta2, double eps,
                                         double max_norm, float* workspace, float* norm_out, parlhip_stream_t stream) {
  if (n_tensors == 0) return PARLHIP_OK;
  if (n_tensors < 0 || n_tensors > kOptMaxTensors) return PARLHIP_ENOSUP;
  if (!params || !grads || !exp_avg || !exp_avg_sq || !steps || !numel || !lr || !workspace) return PARLHIP_EINVAL;
  if (!(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0) || !(max_norm > 0.0))
    return PARLHIP_EINVAL;
  OptTable t;
  t.n = n_tensors;
  int b = 0;
  for (int i = 0; i < kOptMaxTensors; ++i) {
    const bool live = i < n_tensors;
    if (live && (numel[i] <= 0 || !params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i] || !steps[i]))
      return PARLHIP_EINVAL;
    t.p[i] = live ? params[i] : nullptr;
    t.g[i] = live ? grads[i] : nullptr;
    t.m[i] = live ? exp_avg[i] : nullptr;
    t.v[i] = live ? exp_avg_sq[i] : nullptr;
    t.step[i] = live ? steps[i] : nullptr;
    t.numel[i] = live ? (long long)numel[i] : 0;
    t.first_block[i] = b;
    if (live) {
      const long long nb = (numel[i] + kOptChunk - 1) / kOptChunk;
      if (nb + b > 0x7fffffffLL) return PARLHIP_EINVAL;
      b += (int)nb;
    }
  }
  t.first_block[kOptMaxTensors] = b;
  hipStream_t s = (hipStream_t)stream;
  grad_sqsum_kernel<<<b, 256, 0, s>>>(t, workspace);
  int rc = check_launch();
  if (rc) return rc;
  clip_adam_kernel<<<b, 256, 0, s>>>(t, workspace, b, lr, beta1, beta2, (float)eps, (float)max_norm, norm_out);
  return check_launch();
}
