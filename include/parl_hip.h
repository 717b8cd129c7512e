/*
 * parl_hip.h — C ABI of libparl_hip.so, the MI355X (gfx950) hot path behind PARL's
 * IMPALA / A2C / PPO actor-learner API.
 *
 * PARL itself has no C ABI or FFI: its boundary for this path is Python duck typing
 * (SURVEY.md §8b).  The entry points below are what a ctypes binding in the reference
 * would call instead of the Python/numpy/paddle code cited on each one (paths relative
 * to the reference tree).  INTEGRATION.md shows the reference-side stubs.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (e.g. a torch tensor's
 *    data_ptr()), contiguous, in the layout documented on the entry point;
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is
 *    enqueued asynchronously on it, nothing synchronises;
 *  - no hidden allocation: ops that need scratch take a caller-provided workspace whose
 *    size is returned by the matching *_workspace_bytes() query;
 *  - return value: 0 on success, a negative PARLHIP_E* code otherwise (no exceptions
 *    cross the boundary); parlhip_strerror() describes a code;
 *  - thread-safe across distinct streams.
 */
#ifndef PARL_HIP_H_
#define PARL_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PARLHIP_OK 0
#define PARLHIP_EINVAL (-1)   /* bad argument (null pointer, negative size, bad flag) */
#define PARLHIP_ELAUNCH (-2)  /* hipLaunchKernel / runtime error; see parlhip_last_hip_error */
#define PARLHIP_ENOSUP (-3)   /* combination not supported by this build */
#define PARLHIP_ENOMEM (-4)   /* workspace too small */

typedef void* parlhip_stream_t;

/* library version (major*10000 + minor*100 + patch) and error text */
int parlhip_version(void);
const char* parlhip_strerror(int code);
/* hipError_t (as int) of the last failing runtime call on this thread, 0 if none */
int parlhip_last_hip_error(void);
/* Kernels cannot return codes; a kernel that meets bad DATA (an action index outside
 * [0,A)) clamps it and raises a device-side flag.  This call SYNCHRONISES `stream`, returns
 * the flag (0 = clean, >0 = data error seen since the last call, <0 = PARLHIP_E*) and
 * clears it.  Meant for tests and debug builds of the host code, not the steady state.   */
int parlhip_consume_device_errors(parlhip_stream_t stream);

/* ------------------------------------------------------------------------------------
 * V-trace
 * ------------------------------------------------------------------------------------ */

/* vtrace.from_importance_weights — parl/algorithms/paddle/impala/vtrace.py:36-139.
 * All inputs [T,B] float32 time-major (B contiguous); bootstrap_value [B].
 * Outputs vs, pg_advantages [T,B].
 * clip_rho_threshold / clip_pg_rho_threshold: a NaN disables that clip (the reference's
 * `None`, vtrace.py:102-105,131-134).  cs is always min(rho, 1.0) (vtrace.py:107).     */
int parlhip_vtrace_f32(const float* behaviour_actions_log_probs,
                       const float* target_actions_log_probs,
                       const float* discounts, const float* rewards,
                       const float* values, const float* bootstrap_value,
                       float* vs, float* pg_advantages, int T, int B,
                       float clip_rho_threshold, float clip_pg_rho_threshold,
                       parlhip_stream_t stream);

/* Fused learner pre-processing + V-trace: IMPALA._log_prob for both policies
 * (impala.py:119-132), discounts = (~dones)*gamma (impala.py:59), the drop-last-step /
 * bootstrap slicing (impala.py:186-194) and from_importance_weights, in one pass.
 *
 * Inputs cover the FULL rollout of T steps per sequence (T = sample_batch_steps):
 *   behaviour_logits, target_logits : float32 [T,B,A] (time_major=1) or [B,T,A] (=0,
 *                                     the reference's flat env-major batch, impala.py:167-175)
 *   actions : int64 [T,B] / [B,T];  rewards : float32;  dones : uint8 (bool);
 *   values  : float32 (value head output for every step; step T-1 is the bootstrap).
 * Outputs hold T-1 transitions, in the SAME major order as the inputs:
 *   vs, pg_advantages, target_action_log_probs (optional, may be NULL),
 *   behaviour_action_log_probs (optional): [T-1,B] or [B,T-1].                          */
int parlhip_vtrace_from_logits_f32(const float* behaviour_logits,
                                   const float* target_logits,
                                   const int64_t* actions, const float* rewards,
                                   const uint8_t* dones, const float* values,
                                   float* vs, float* pg_advantages,
                                   float* target_action_log_probs,
                                   float* behaviour_action_log_probs, int T, int B,
                                   int A, int time_major, float gamma,
                                   float clip_rho_threshold,
                                   float clip_pg_rho_threshold,
                                   parlhip_stream_t stream);

/* ------------------------------------------------------------------------------------
 * GAE / n-step returns / discounted sums
 * ------------------------------------------------------------------------------------ */

#define PARLHIP_GAE_DONE_ENDS_STEP 0  /* A2C/IMPALA convention: dones[t] = transition t ended
                                         the episode (examples/A2C/actor.py:73-85)          */
#define PARLHIP_GAE_DONE_STARTS_STEP 1 /* PPO RolloutStorage convention: dones[t] = obs t is
                                          the first of a new episode; nextnonterminal =
                                          1-dones[t+1], last step uses last_done
                                          (examples/PPO/storage.py:45-64)                   */

/* Batched calc_gae (parl/utils/rl_utils.py:34-51) with the segment semantics of
 * examples/A2C/actor.py:73-85, or RolloutStorage.compute_returns (examples/PPO/storage.py:45-64).
 * rewards, values: float32 [T,B] time-major.  dones: [T,B], uint8 when dones_are_f32==0,
 * float32 otherwise (PPO storage layout).  next_value: [B] value of the state after the
 * last step (ignored for a sequence whose last transition is terminal).  last_done: [B]
 * (same dtype as dones), only read in PARLHIP_GAE_DONE_STARTS_STEP mode.
 * Outputs: advantages [T,B]; returns [T,B] = advantages + values (A2C `target_values`,
 * PPO `returns`); either may be NULL.  lambda == 1 gives the n-step return.             */
int parlhip_gae_f32(const float* rewards, const float* values, const void* dones,
                    const float* next_value, const void* last_done,
                    float* advantages, float* returns, int T, int B, float gamma,
                    float lam, int done_convention, int dones_are_f32,
                    parlhip_stream_t stream);

/* calc_discount_sum_rewards (parl/utils/rl_utils.py:21-31), batched: x [T,B] float32,
 * out[t] = x[t] + gamma*out[t+1]; optional uint8 dones [T,B] reset the carry after a
 * terminal step (NULL = plain lfilter semantics).                                       */
int parlhip_discount_cumsum_f32(const float* x, const uint8_t* dones, float* out,
                                int T, int B, float gamma, parlhip_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Advantage normalisation (PPO minibatch): (adv-mean)/(std+eps), UNBIASED std
 * parl/algorithms/paddle/ppo.py:124-127, parl/algorithms/torch/ppo.py:115-117
 * ------------------------------------------------------------------------------------ */
size_t parlhip_adv_normalize_workspace_bytes(int64_t n);
/* adv: float32 [n_total]; idx: optional int64 [n] gather indices into adv (the shuffled
 * minibatch of examples/PPO/agent.py:91-110), NULL = adv[0..n).  out: float32 [n].
 * mean_std_out: optional float32 [2].                                                  */
int parlhip_adv_normalize_f32(const float* adv, const int64_t* idx, float* out,
                              int64_t n, float eps, void* workspace,
                              size_t workspace_bytes, float* mean_std_out,
                              parlhip_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Categorical action sampling — np.random.choice(len(prob), 1, p=prob) per row,
 * examples/IMPALA/atari_agent.py:38-40, examples/A2C/atari_agent.py:52-54.
 * ------------------------------------------------------------------------------------ */

/* probs float32 [B,A]; uniforms float64 [B] in [0,1).  actions int64 [B] =
 * searchsorted(cumsum_f64(probs)/sum, u, side='right') (numpy legacy choice).           */
int parlhip_categorical_sample_f32(const float* probs, const double* uniforms,
                                   int64_t* actions, int B, int A,
                                   parlhip_stream_t stream);

/* Same with on-device uniforms: u[b] = philox4x32-10(key=seed, counter=(offset, row0+b))
 * mapped to a 53-bit double.  Optionally emits the uniforms (uniforms_out, may be NULL).
 * logits_or_probs: if is_logits, probs = softmax_f32(logits) (IMPALA.sample,
 * impala.py:217-227) and probs_out (optional) receives them.                            */
int parlhip_policy_sample_f32(const float* logits_or_probs, int is_logits,
                              int64_t* actions, float* probs_out,
                              double* uniforms_out, int B, int A, uint64_t seed,
                              uint64_t offset, uint64_t row0, parlhip_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PARL_HIP_H_ */
