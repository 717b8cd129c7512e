// policy_head.hpp — the actors' policy head + draw for ONE env row on one wavefront, shared by
// policy_head_sample_kernel (sample_norm_kernels.hip) and the head of atari_env_kernel
// (parlhip_atari_vec_step_policy_obs): the same instructions in both, so the two forms draw the same actions
// from the same logits bit for bit.
//
// Reference: examples/IMPALA/atari_model.py:44-57 (policy_fc), parl/algorithms/paddle/impala/impala.py:217-227
// (IMPALA.sample: softmax of the logits), examples/IMPALA/atari_agent.py:35-42 (np.random.choice per env):
// lane l holds h[row][4l .. 4l+3]; a logit is four FMAs per lane and a wave reduction plus the bias; the draw is
// float32 softmax, float64 inverse CDF on the Philox uniform of (offset, row) — policy_sample_kernel's arithmetic.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "philox.hpp"

namespace parlhip {

__device__ __forceinline__ float wave_sum_f32(float x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
  return x;
}

// logits of the wave's row into row[0 .. A) — every lane ends with all of them
template <int A_MAX>
__device__ __forceinline__ void policy_head_row(const float* __restrict__ h_row, const float* __restrict__ w,
                                                const float* __restrict__ bias, int A, int lane, float (&row)[A_MAX]) {
  const float4 hv = ((const float4*)h_row)[lane];
#pragma unroll
  for (int k = 0; k < A_MAX; ++k) {
    row[k] = 0.f;
    if (k < A) {
      const float4 wv = ((const float4*)(w + (size_t)k * 256))[lane];
      float p = hv.x * wv.x;
      p = __builtin_fmaf(hv.y, wv.y, p);
      p = __builtin_fmaf(hv.z, wv.z, p);
      p = __builtin_fmaf(hv.w, wv.w, p);
      row[k] = wave_sum_f32(p) + bias[k];
    }
  }
}

// the action drawn from softmax(row) with uniform u (searchsorted(cumsum_f64(p) / cdf[-1], u, side='right'))
template <int A_MAX>
__device__ __forceinline__ int64_t policy_draw(const float (&row)[A_MAX], int A, double u) {
  float m = row[0];
#pragma unroll
  for (int k = 1; k < A_MAX; ++k) if (k < A) m = fmaxf(m, row[k]);
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < A_MAX; ++k) if (k < A) s += expf(row[k] - m);
  double last = 0.0;
#pragma unroll
  for (int k = 0; k < A_MAX; ++k) if (k < A) last += (double)(expf(row[k] - m) / s);
  double c = 0.0;
  int64_t a = A;
#pragma unroll
  for (int k = 0; k < A_MAX; ++k) {
    if (k < A) {
      c += (double)(expf(row[k] - m) / s);
      if (a == A && c / last > u) a = k;
    }
  }
  return a;
}

}  // namespace parlhip
