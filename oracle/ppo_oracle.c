/*
 * ppo_oracle.c — CPU restatement (plain C, float64) of the PPO example's running normalisation
 * and minibatch gather.  TEST INFRASTRUCTURE ONLY (see scan_oracle.c header).
 *
 * Pinning: tests/test_oracle_golden.py checks these against outputs of the reference's own
 * VecNormalizeEnv / RunningMeanStd / RolloutStorage classes run in the build container
 * (tests/golden/make_ppo_golden.py -> vecnormalize.npz, ppo_sample_batch.npz), bit-exact.
 * Compile with -ffp-contract=off (numpy does not fuse multiply-adds).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

/* parl/env/mujoco_wrappers.py:185-206 update_mean_var_count_from_moments with
 * batch_mean = x, batch_var = 0, batch_count = 1 (RunningMeanStd.update on a batch of one,
 * :83-87: np.mean / np.var over one row are x and 0 exactly)                                */
static void rms_update1(double x, double count, double* mean, double* var) {
  const double delta = x - *mean;                          /* :196 */
  const double tot = count + 1.0;                          /* :197 */
  const double new_mean = *mean + (delta * 1.0) / tot;     /* :199 */
  const double m_a = *var * count;                         /* :200 */
  const double m_b = 0.0 * 1.0;                            /* :201 */
  const double M2 = (m_a + m_b) + (((delta * delta) * count) * 1.0) / tot; /* :202 */
  *mean = new_mean;
  *var = M2 / tot;                                         /* :203 */
}

static double clipd(double v, double lo, double hi) {
  v = v < lo ? lo : v;
  return v > hi ? hi : v;
}

/* VecNormalizeEnv._obfilt, mujoco_wrappers.py:140-156, one RunningMeanStd per env */
int oracle_vecnorm_obs_f64(const double* raw, double* mean, double* var, double* count,
                           const uint8_t* mask, float* out, double* out64, int E, int D,
                           double clipob, double eps, int update) {
  if (E < 0 || D < 0) return -1;
  for (int e = 0; e < E; ++e) {
    if (mask && !mask[e]) continue;
    const double cnt = count[e];
    for (int d = 0; d < D; ++d) {
      const size_t i = (size_t)e * (size_t)D + (size_t)d;
      if (update) rms_update1(raw[i], cnt, &mean[i], &var[i]);                  /* :147-148 */
      const double o = clipd((raw[i] - mean[i]) / sqrt(var[i] + eps), -clipob, clipob); /* :149-151 */
      if (out) out[i] = (float)o;
      if (out64) out64[i] = o;
    }
    if (update) count[e] = cnt + 1.0;
  }
  return 0;
}

/* VecNormalizeEnv.step, mujoco_wrappers.py:120-136 (reward half) */
int oracle_vecnorm_reward_f64(const double* rew, const uint8_t* done, double* ret, double* ret_mean,
                              double* ret_var, double* ret_count, float* out, double* out64, int E,
                              double gamma, double cliprew, double eps) {
  if (E < 0) return -1;
  for (int e = 0; e < E; ++e) {
    const double acc = ret[e] * gamma + rew[e];            /* :122 */
    rms_update1(acc, ret_count[e], &ret_mean[e], &ret_var[e]);  /* :127 */
    ret_count[e] = ret_count[e] + 1.0;
    const double o = clipd(rew[e] / sqrt(ret_var[e] + eps), -cliprew, cliprew); /* :128-129 */
    ret[e] = done[e] ? 0.0 : acc;                          /* :131-132 */
    if (out) out[e] = (float)o;
    if (out64) out64[e] = o;
  }
  return 0;
}

/* RolloutStorage.sample_batch, examples/PPO/storage.py:66-76 */
int oracle_ppo_sample_batch_f32(const float* obs, const float* actions, const float* logprobs,
                                const float* advantages, const float* returns, const float* values,
                                const int64_t* idx, float* o_obs, float* o_act, float* o_logp,
                                float* o_adv, float* o_ret, float* o_val, int64_t N, int64_t M,
                                int obs_dim, int act_dim) {
  for (int64_t m = 0; m < M; ++m) {
    const int64_t j = idx[m];
    if (j < 0 || j >= N) return -2;
    for (int d = 0; d < obs_dim; ++d) o_obs[m * obs_dim + d] = obs[j * obs_dim + d];
    for (int d = 0; d < act_dim; ++d) o_act[m * act_dim + d] = actions[j * act_dim + d];
    o_logp[m] = logprobs[j];
    o_adv[m] = advantages[j];
    o_ret[m] = returns[j];
    o_val[m] = values[j];
  }
  return 0;
}
