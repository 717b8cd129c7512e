// ppo_kernels.hip — the PPO rows of SURVEY.md §8 (a10 RolloutStorage.sample_batch, f4
// VecNormalizeEnv): host-stepped environments (MuJoCo) feed raw float64 observations / rewards,
// everything after that stays on the GPU.
//
// Reference: parl/env/mujoco_wrappers.py:73-206 (RunningMeanStd, VecNormalizeEnv,
// update_mean_var_count_from_moments) — the reference runs ONE VecNormalizeEnv per environment
// (examples/PPO/env_utils.py:118-127), i.e. every update is a batch of one sample
// (batch_mean = x, batch_var = 0, batch_count = 1), in float64 numpy.  The kernels keep numpy's
// operation order in float64 (-ffp-contract=off, IEEE divide / sqrt), so the running statistics
// and the normalised outputs are bit-identical to the reference's; the float32 output is the
// cast RolloutStorage.append's float32 assignment performs (examples/PPO/storage.py:35-41).
// examples/PPO/storage.py:66-76 + examples/PPO/agent.py:91-110: minibatch gather.
#include "common.hpp"

namespace parlhip {

// update_mean_var_count_from_moments (mujoco_wrappers.py:185-206) for batch_mean = x,
// batch_var = 0, batch_count = 1
__device__ __forceinline__ void rms_update1(double x, double count, double& mean, double& var) {
  const double delta = x - mean;
  const double tot = count + 1.0;
  const double new_mean = mean + (delta * 1.0) / tot;
  const double m_a = var * count;
  const double m_b = 0.0 * 1.0;
  const double M2 = (m_a + m_b) + (((delta * delta) * count) * 1.0) / tot;
  mean = new_mean;
  var = M2 / tot;
}

__device__ __forceinline__ double clip(double v, double lo, double hi) {
  v = v < lo ? lo : v;   // np.clip = minimum(maximum(v, lo), hi)
  return v > hi ? hi : v;
}

// one wavefront per environment: its D statistics are touched by this wave only, so the shared
// per-env count is read before and written after without a cross-wave hazard
__global__ __launch_bounds__(256) void vecnorm_obs_kernel(
    const double* __restrict__ raw, double* __restrict__ mean, double* __restrict__ var,
    double* __restrict__ count, const uint8_t* __restrict__ mask, float* __restrict__ out,
    double* __restrict__ out64, int E, int D, double clipob, double eps, int update) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int e = wave; e < E; e += nwaves) {
    if (mask && !mask[e]) continue;
    const double cnt = count[e];
    for (int d = lane; d < D; d += 64) {
      const size_t i = (size_t)e * D + d;
      const double x = raw[i];
      double m = mean[i], v = var[i];
      if (update) {
        rms_update1(x, cnt, m, v);
        mean[i] = m;
        var[i] = v;
      }
      const double o = clip((x - m) / sqrt(v + eps), -clipob, clipob);
      if (out) out[i] = (float)o;
      if (out64) out64[i] = o;
    }
    if (update && lane == 0) count[e] = cnt + 1.0;
  }
}

__global__ __launch_bounds__(256) void vecnorm_reward_kernel(
    const double* __restrict__ rew, const uint8_t* __restrict__ done, double* __restrict__ ret,
    double* __restrict__ ret_mean, double* __restrict__ ret_var, double* __restrict__ ret_count,
    float* __restrict__ out, double* __restrict__ out64, int E, double gamma, double cliprew,
    double eps) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const double r = rew[e];
  const double acc = ret[e] * gamma + r;               // mujoco_wrappers.py:122
  double m = ret_mean[e], v = ret_var[e];
  const double cnt = ret_count[e];
  rms_update1(acc, cnt, m, v);                          // :127 ret_rms.update(self.ret)
  ret_mean[e] = m;
  ret_var[e] = v;
  ret_count[e] = cnt + 1.0;
  const double o = clip(r / sqrt(v + eps), -cliprew, cliprew);   // :128-129
  ret[e] = done[e] ? 0.0 : acc;                         // :131-132
  if (out) out[e] = (float)o;
  if (out64) out64[e] = o;
}

// RolloutStorage.sample_batch (storage.py:66-76): six gathers by the same minibatch index in one
// launch.  One wavefront per minibatch row: the obs / action rows are contiguous (coalesced), the
// four scalars ride along on lanes 0-3.
__global__ __launch_bounds__(256) void ppo_sample_batch_kernel(
    const float* __restrict__ obs, const float* __restrict__ act, const float* __restrict__ logp,
    const float* __restrict__ adv, const float* __restrict__ ret, const float* __restrict__ val,
    const int64_t* __restrict__ idx, float* __restrict__ o_obs, float* __restrict__ o_act,
    float* __restrict__ o_logp, float* __restrict__ o_adv, float* __restrict__ o_ret,
    float* __restrict__ o_val, int64_t N, int64_t M, int Do, int Da, int* __restrict__ err) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t m = wave; m < M; m += nwaves) {
    const int64_t j = idx[m];
    if (j < 0 || j >= N) {  // numpy would raise IndexError: flag it, write nothing
      if (lane == 0) atomicOr(err, 1);
      continue;
    }
    for (int d = lane; d < Do; d += 64) o_obs[m * Do + d] = obs[j * Do + d];
    for (int d = lane; d < Da; d += 64) o_act[m * Da + d] = act[j * Da + d];
    if (lane == 0) o_logp[m] = logp[j];
    if (lane == 1) o_adv[m] = adv[j];
    if (lane == 2) o_ret[m] = ret[j];
    if (lane == 3) o_val[m] = val[j];
  }
}

int* device_error_flag();  // scan_kernels.hip

}  // namespace parlhip

using namespace parlhip;

PARLHIP_EXPORT int parlhip_vecnorm_obs_f64(const double* raw, double* mean, double* var, double* count,
                                           const uint8_t* mask, float* out, double* out64, int E, int D,
                                           double clipob, double eps, int update, parlhip_stream_t stream) {
  if (E < 0 || D < 0) return PARLHIP_EINVAL;
  if (E == 0 || D == 0) return PARLHIP_OK;
  if (!raw || !mean || !var || !count || (!out && !out64)) return PARLHIP_EINVAL;
  const int waves_per_block = 4;
  int grid = ceil_div(E, waves_per_block);
  if (grid > 16 * kNumCU) grid = 16 * kNumCU;
  vecnorm_obs_kernel<<<grid, 64 * waves_per_block, 0, (hipStream_t)stream>>>(raw, mean, var, count, mask, out, out64,
                                                                               E, D, clipob, eps, update);
  return check_launch();
}

PARLHIP_EXPORT int parlhip_vecnorm_reward_f64(const double* rew, const uint8_t* done, double* ret,
                                              double* ret_mean, double* ret_var, double* ret_count, float* out,
                                              double* out64, int E, double gamma, double cliprew, double eps,
                                              parlhip_stream_t stream) {
  if (E < 0) return PARLHIP_EINVAL;
  if (E == 0) return PARLHIP_OK;
  if (!rew || !done || !ret || !ret_mean || !ret_var || !ret_count || (!out && !out64)) return PARLHIP_EINVAL;
  vecnorm_reward_kernel<<<ceil_div(E, 256), 256, 0, (hipStream_t)stream>>>(rew, done, ret, ret_mean, ret_var, ret_count,
                                                                           out, out64, E, gamma, cliprew, eps);
  return check_launch();
}

PARLHIP_EXPORT int parlhip_ppo_sample_batch_f32(const float* obs, const float* actions, const float* logprobs,
                                                const float* advantages, const float* returns, const float* values,
                                                const int64_t* idx, float* out_obs, float* out_actions,
                                                float* out_logprobs, float* out_advantages, float* out_returns,
                                                float* out_values, int64_t N, int64_t M, int obs_dim, int act_dim,
                                                parlhip_stream_t stream) {
  if (N < 0 || M < 0 || obs_dim < 0 || act_dim < 0) return PARLHIP_EINVAL;
  if (M == 0) return PARLHIP_OK;
  if (!idx || !logprobs || !advantages || !returns || !values || !out_logprobs || !out_advantages || !out_returns ||
      !out_values || (obs_dim && (!obs || !out_obs)) || (act_dim && (!actions || !out_actions)))
    return PARLHIP_EINVAL;
  int64_t grid = (M + 3) / 4;
  if (grid > 32 * kNumCU) grid = 32 * kNumCU;
  ppo_sample_batch_kernel<<<(int)grid, 256, 0, (hipStream_t)stream>>>(
      obs, actions, logprobs, advantages, returns, values, idx, out_obs, out_actions, out_logprobs, out_advantages,
      out_returns, out_values, N, M, obs_dim, act_dim, device_error_flag());
  return check_launch();
}
