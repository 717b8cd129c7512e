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
/* sha256[:16] over the sources the library was built from (csrc/srchash.py): build hygiene, lets a
 * test notice a library that does not match the tree */
const char* parlhip_source_hash(void);
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

/* IMPALA learner loss in one pass (impala.py:25-79 VTraceLoss + impala.py:119-194 of IMPALA.learn):
 * everything parlhip_vtrace_from_logits_f32 does, plus the Categorical entropy / KL of the two
 * policies, the three loss sums and the gradient of
 *     total = pi_loss + vf_coeff * vf_loss + ent_coeff * entropy          (impala.py:78-79)
 * with respect to target_logits and values (the V-trace targets carry no gradient, vtrace.py:36).
 * Inputs as parlhip_vtrace_from_logits_f32.  Outputs: vs, pg_advantages [T-1,B] / [B,T-1];
 * grad_logits f32 in the layout of target_logits, grad_values f32 in the layout of values (rows of
 * the bootstrap step are zero); sums f64[4], ADDED to (zero them first): pi_loss, vf_loss,
 * entropy, and the sum over all T*B rows of KL(target || behaviour) (impala.py:161-165 takes
 * its mean).  Returns PARLHIP_ENOSUP for T > 256 or an action count without a compiled
 * instantiation (2, 3, 4, 6, 9, 18): callers then use the unfused entry.                          */
int parlhip_impala_loss_f32(const float* behaviour_logits, const float* target_logits,
                            const int64_t* actions, const float* rewards, const uint8_t* dones,
                            const float* values, float* vs, float* pg_advantages, float* grad_logits,
                            float* grad_values, double* sums, int T, int B, int A, int time_major,
                            float gamma, float clip_rho_threshold, float clip_pg_rho_threshold,
                            float vf_coeff, float entropy_coeff, parlhip_stream_t stream);

/* parlhip_impala_loss_f32 with the two heads of the network in front and behind it: IMPALA.learn's
 * policy_fc / value_fc (examples/IMPALA/atari_model.py:44-57, :73-90), the loss, and the heads'
 * backward in one kernel.  hidden f32 [T,B,H] (the trunk's output, time-major rows), w_policy [A,H],
 * b_policy [A], w_value [H] (= value_fc.weight [1,H]), b_value [1]; the loss inputs as above
 * (time-major).  Outputs: vs, pg_advantages [T-1,B]; grad_hidden [T,B,H] = d total / d hidden;
 * grad_heads f32 [(A+1)*H + (A+1)]: d total / d w_policy rows, d / d w_value, then d / d b_policy,
 * d / d b_value; sums as above (ADDED to).  workspace: parlhip_impala_heads_loss_workspace_bytes(B, A)
 * bytes, 16-byte aligned like hidden / grad_hidden / the weights.  The trunk output and its gradient
 * cross HBM once each (the four head GEMMs of the framework path read / write them 3 times).
 * PARLHIP_ENOSUP unless H == 256, T <= 64, A in {4, 6}: callers use parlhip_impala_loss_f32.      */
size_t parlhip_impala_heads_loss_workspace_bytes(int B, int A);
int parlhip_impala_heads_loss_f32(const float* hidden, const float* w_policy, const float* b_policy,
                                  const float* w_value, const float* b_value,
                                  const float* behaviour_logits, const int64_t* actions,
                                  const float* rewards, const uint8_t* dones, float* vs,
                                  float* pg_advantages, float* grad_hidden, float* grad_heads,
                                  double* sums, void* workspace, int T, int B, int hidden_units, int A,
                                  float gamma, float clip_rho_threshold, float clip_pg_rho_threshold,
                                  float vf_coeff, float entropy_coeff, parlhip_stream_t stream);

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

/* Same arithmetic with a caller workspace, for long rollouts over few sequences (PPO: T=2048,
 * E=4096, examples/PPO/storage.py:45-64): T is cut into 32-step chunks that run in parallel in ONE
 * pass over HBM (affine recurrence: every chunk keeps its steps in registers, publishes its
 * aggregate, folds the aggregates of the later chunks and finishes with the exact-order recurrence
 * inside the chunk).  The workspace (8 B per chunk and sequence, 16-byte aligned) is overwritten.
 * parlhip_gae_workspace_bytes() returns 0 when the single-pass
 * kernel is the better plan for (T, B); parlhip_gae_ws_f32 then forwards to parlhip_gae_f32 and
 * workspace may be NULL.  Results agree with parlhip_gae_f32 to fp32 re-association (<=1e-6 rel). */
size_t parlhip_gae_workspace_bytes(int T, int B);
int parlhip_gae_ws_f32(const float* rewards, const float* values, const void* dones,
                       const float* next_value, const void* last_done, float* advantages,
                       float* returns, int T, int B, float gamma, float lam, int done_convention,
                       int dones_are_f32, void* workspace, size_t workspace_bytes,
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
/* The same with the Philox offset = offset + *offset_base, offset_base a uint64 in device memory: a rollout replayed
 * as a hipGraph (frozen kernel arguments) keeps the number of its first step there.                        */
int parlhip_policy_sample_at_f32(const float* x, int is_logits, int64_t* actions, float* probs_out,
                                 double* uniforms_out, int B, int A, uint64_t seed,
                                 const uint64_t* offset_base, uint64_t offset, uint64_t row0,
                                 parlhip_stream_t stream);

/* The actors' policy head and the draw in ONE launch: logits = hidden [B,256] @ w_policy^T [A,256] + b_policy
 * (examples/IMPALA/atari_model.py:44-57,73-79) written to logits_out [B,A] (a rollout slab), then the action
 * exactly as parlhip_policy_sample_f32(is_logits = 1) draws it.  hidden_units must be 256, A <= 18.       */
int parlhip_policy_head_sample_f32(const float* hidden, const float* w_policy, const float* b_policy,
                                   float* logits_out, int64_t* actions, int B, int hidden_units, int A,
                                   uint64_t seed, uint64_t offset, uint64_t row0, parlhip_stream_t stream);

/* The same launch with the Philox offset = *offset_base (a uint64 in DEVICE memory) + offset: the actors' whole
 * rollout (examples/IMPALA/actor.py:58-76, T env steps) is replayed as a hipGraph whose kernel arguments are
 * frozen — the number of the rollout's first step is the one thing that changes between replays, so it is read
 * from memory; `offset` is then the step's index inside the rollout.  Draws the same actions as the entry above
 * called with offset = *offset_base + offset.                                                             */
int parlhip_policy_head_sample_at_f32(const float* hidden, const float* w_policy, const float* b_policy,
                                      float* logits_out, int64_t* actions, int B, int hidden_units, int A,
                                      uint64_t seed, const uint64_t* offset_base, uint64_t offset, uint64_t row0,
                                      parlhip_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Frame pipeline: MaxAndSkipEnv max + WarpFrame
 * parl/env/atari_wrappers.py:239 (obs_buffer.max(axis=0)), :263-267 (cv2 RGB2GRAY +
 * cv2.resize INTER_AREA), restated from OpenCV's published algorithm (oracle/frame_oracle.c).
 * ------------------------------------------------------------------------------------ */
/* Resampling tables + NTSC palette: built on the HOST into a caller buffer of
 * parlhip_frame_post_tables_bytes(dim) bytes, then copied to the device by the caller. */
size_t parlhip_frame_post_tables_bytes(int dim);
int parlhip_frame_post_tables_init(void* host_blob, int dim);
/* frames0/frames1: E frames, in_stride bytes apart.  fmt 0: RGB u8 [210,160,3] (the WarpFrame
 * boundary); fmt 1: TIA colour bytes [210,160] as the device emulator emits them (16-byte
 * aligned).  frames1 NULL = no max.  flags (optional, [E]): bit0 set = ignore frames1 for that
 * env.  out: E frames of dim*dim bytes, out_stride bytes apart (e.g. a rollout-ring slot).   */
int parlhip_frame_post_u8(const uint8_t* frames0, const uint8_t* frames1, int64_t in_stride,
                          int fmt, const uint8_t* flags, uint8_t* out, int64_t out_stride,
                          int E, int dim, const void* tables_dev, parlhip_stream_t stream);
/* The same plus the FrameStack bookkeeping of parlhip_stack_since_update_u8 in ONE launch (the device
 * env's per-step path): since_next[e] = (flags[e] & 2) ? 0 : min(since_prev[e] + 1, 3); since_prev
 * may be NULL (treated as 0).  dim <= 84.                                                        */
int parlhip_frame_post_since_u8(const uint8_t* frames0, const uint8_t* frames1, int64_t in_stride, int fmt,
                                const uint8_t* flags, uint8_t* out, int64_t out_stride, int E, int dim,
                                const void* tables_dev, const uint8_t* since_prev, uint8_t* since_next,
                                parlhip_stream_t stream);
/* The same plus the MonitorEnv statistics of parlhip_episode_stats_accum_f64 (atari_wrappers.py:88-95) in the
 * same launch: the whole post-emulator part of VectorEnv.step for E envs is ONE kernel.              */
int parlhip_frame_post_step_u8(const uint8_t* frames0, const uint8_t* frames1, int64_t in_stride, int fmt,
                               const uint8_t* flags, uint8_t* out, int64_t out_stride, int E, int dim,
                               const void* tables_dev, const uint8_t* since_prev, uint8_t* since_next,
                               const float* ep_returns, const int32_t* ep_lengths, double* ep_acc3,
                               parlhip_stream_t stream);


/* ------------------------------------------------------------------------------------
 * Vectorised Atari env: VectorEnv([wrap_deepmind(gym.make(id), dim, obs_format='NCHW')]*E)
 * parl/env/vector_env.py:34-63 + parl/env/atari_wrappers.py:356-385 + the ALE emulator
 * behind gym.make (third party; restated, see oracle/atari_oracle.h).  One env per wavefront for the 6507 / RIOT /
 * wrapper chain, plus (round 4) a second wavefront of the same workgroup that draws the env's picture from the
 * stream of TIA register records the first one produces.
 * ------------------------------------------------------------------------------------ */
#define PARLHIP_GAME_PONG 1
#define PARLHIP_GAME_BREAKOUT 2
size_t parlhip_atari_state_bytes(void);       /* per-env state blob (device), E of them      */
size_t parlhip_atari_frame_bytes(void);       /* per-env raw frame pair (device): 2*210*160  */
size_t parlhip_atari_rom_table_bytes(uint32_t rom_size);
size_t parlhip_atari_reset_cache_bytes(void); /* 30 reset snapshots (device)                 */
int parlhip_atari_num_actions(int game);      /* ALE minimal action set size                 */
/* HOST: pre-decode an unbanked 2K/4K cartridge into one 32-bit word per address.  When the     */
/* library carries natively translated code for exactly this cartridge (CRC-32 match, see        */
/* parlhip_atari_native_cart) the table is tagged and the env kernel runs the translated code,   */
/* otherwise it interprets; results are identical either way.                                    */
int parlhip_atari_rom_table_build(const uint8_t* rom_host, uint32_t rom_size,
                                  uint32_t* table_host);
/* CRC-32 of the cartridge whose program was statically translated to gfx950 code for `game`     */
/* when the library was built (csrc/gen_cart_native.py; stands for the ALE core's interpreter    */
/* loop behind gym.make, examples/IMPALA/actor.py:34), or 0 if the library only interprets.      */
uint32_t parlhip_atari_native_cart(int game);
/* Build the 30 real-reset snapshots (noop count 1..30) on the device.  jam_flag_dev: int32
 * word OR-ed with emulator fault bits (undocumented opcode etc.); 0 = clean.                 */
int parlhip_atari_reset_cache_build(const uint32_t* rom_table_dev, uint32_t rom_size, int game,
                                    int64_t max_episode_steps, void* cache_dev,
                                    int32_t* jam_flag_dev, parlhip_stream_t stream);
/* VectorEnv.reset: initialises states; leaves each env's reset frames in `frames` and
 * obs_flags[e] = 2 | single (bit1: the frame stack must be refilled, FrameStack.reset).      */
int parlhip_atari_vec_reset(void* states, const uint32_t* rom_table_dev, uint32_t rom_size,
                            int game, uint8_t* frames, uint8_t* obs_flags, int E, uint64_t seed,
                            uint64_t env_id0, int64_t max_episode_steps, int32_t* jam_flag_dev,
                            parlhip_stream_t stream);
/* VectorEnv.step with auto-reset.  actions int64 [E] index the minimal action set.  Outputs:
 * rewards f32 [E] (ClipRewardEnv sign), dones u8 [E], obs_flags u8 [E], the episode that
 * MonitorEnv closed this step if any (ep_lengths[e] > 0; lengths in emulated frames, returns
 * unclipped), and the raw frame pair of the returned observation in `frames` (feed to
 * parlhip_frame_post_u8 with fmt 1 and flags = obs_flags).  reset_cache_dev may be NULL
 * (general reset path only).  env ids env_id0+e select the per-env noop RNG stream.          */
int parlhip_atari_vec_step(void* states, const uint32_t* rom_table_dev, uint32_t rom_size,
                           int game, const int64_t* actions, uint8_t* frames, float* rewards,
                           uint8_t* dones, uint8_t* obs_flags, float* ep_returns,
                           int32_t* ep_lengths, int E, uint64_t seed, uint64_t env_id0,
                           int64_t max_episode_steps, const void* reset_cache_dev,
                           int32_t* jam_flag_dev, parlhip_stream_t stream);
/* VectorEnv.step WITH its observation (parl/env/vector_env.py:41-63 returns the stacked obs of the step; the
 * chain behind it: MaxAndSkipEnv max atari_wrappers.py:239, WarpFrame :263-267, FrameStack :270-306): what
 * parlhip_atari_vec_step + parlhip_frame_post_step_u8(fmt 1, flags = obs_flags) do in two launches, in ONE —
 * each env's two wavefronts convert its frame pair at their tail, as soon as the picture is drawn, instead of a
 * second launch that waits for the slowest env of the grid.  Bit-identical outputs.  obs_out u8 [E, dim*dim] (the
 * ring slot of this step), tables_dev from parlhip_frame_post_tables_init(dim), since_prev (may be NULL) /
 * since_next u8 [E] as in parlhip_frame_post_since_u8, ep_acc3 f64 [3] as in parlhip_episode_stats_accum_f64 (may
 * be NULL).  `frames` still receives the raw pair.  PARLHIP_ENOSUP unless dim is 42 or 84 and the cartridge is
 * an unbanked 2K one (the tail's LDS is the half of the cartridge table a 2K cartridge leaves free): the caller
 * then uses the two-launch form.                                                                       */
int parlhip_atari_vec_step_obs(void* states, const uint32_t* rom_table_dev, uint32_t rom_size, int game,
                               const int64_t* actions, uint8_t* frames, float* rewards, uint8_t* dones,
                               uint8_t* obs_flags, float* ep_returns, int32_t* ep_lengths, int E,
                               uint64_t seed, uint64_t env_id0, int64_t max_episode_steps,
                               const void* reset_cache_dev, int32_t* jam_flag_dev, uint8_t* obs_out, int dim,
                               const void* tables_dev, const uint8_t* since_prev, uint8_t* since_next,
                               double* ep_acc3, parlhip_stream_t stream);
/* The actors' whole per-step tail of examples/IMPALA/actor.py:58-76 in ONE launch: agent.sample's policy head +
 * np.random.choice draw (what parlhip_policy_head_sample_at_f32 does: same instructions, same logits, same
 * actions), vector_env.step, and the observation (parlhip_atari_vec_step_obs).  hidden f32 [E,256] = the trunk
 * output of the actors' model for the observation the envs hold (16-byte aligned), w_policy f32 [A,256], b_policy
 * [A]; logits_out f32 [E,A] and actions_out int64 [E] are this step's rows of the rollout slabs; the draw of env e
 * is the Philox uniform of (sample_seed; offset (+ *offset_base if not NULL), row0 + e).  A must be the game's
 * action count (<= 6), hidden_units 256.  PARLHIP_ENOSUP as parlhip_atari_vec_step_obs.                  */
int parlhip_atari_vec_step_policy_obs(void* states, const uint32_t* rom_table_dev, uint32_t rom_size, int game,
                                      uint8_t* frames, float* rewards, uint8_t* dones, uint8_t* obs_flags,
                                      float* ep_returns, int32_t* ep_lengths, int E, uint64_t seed,
                                      uint64_t env_id0, int64_t max_episode_steps, const void* reset_cache_dev,
                                      int32_t* jam_flag_dev, uint8_t* obs_out, int dim, const void* tables_dev,
                                      const uint8_t* since_prev, uint8_t* since_next, double* ep_acc3,
                                      const float* hidden, const float* w_policy, const float* b_policy,
                                      float* logits_out, int64_t* actions_out, int hidden_units, int A,
                                      uint64_t sample_seed, const uint64_t* offset_base, uint64_t offset,
                                      uint64_t row0, parlhip_stream_t stream);
/* Elastic VectorEnv.step (examples/IMPALA/actor.py:58-76 collects sample_batch_steps steps of every env;
 * the reference's actors are independent processes, so one actor's slow step never holds up another's).
 * A launch emulates at most `frame_budget` (>= 4) frames per env: an env whose step needs more — the 12
 * frames of a life-loss reset (atari_wrappers.py:200-211 + :163-171), a real reset the snapshot cache
 * cannot serve — parks its wrapper state machine in its state blob and goes on in the following
 * launches (taking no action, delivering no observation: obs_flags bit 2 set, parlhip_frame_post_*_u8
 * skips the env) while the other envs keep stepping.  Every env sees exactly the frames and inputs
 * parlhip_atari_vec_step gives it.  rewards / dones / obs_flags / ep_* [E] are per-launch scratch as in
 * parlhip_atari_vec_step (ep_* valid for every launch: feed each to parlhip_episode_stats_accum_f64).
 * Rows: rows_done i32 [E] counts the steps an env has started since the run began; row r of an env
 * lives at index r % rows_ring of the row tables (rows_ring = a multiple of batch_rows, >= 2 batches);
 * an env with rows_done >= rows_limit waits (the caller raises the limit as it consumes batches, so
 * fast envs run ahead into the next batch instead of idling at a batch boundary).
 *   row_launch i32 [rows_ring,E] = `launch` of the call in which the env started the row: the action
 *     and policy output of THAT launch belong to the row, its observation is the one the env held;
 *   rewards_rows f32 / dones_rows u8 [rows_ring,E], written by row; ctl u8 [E] scratch;
 *   finished i32 [2]: finished[m & 1] += 1 when an env starts the last row of batch m.
 * Observations: the caller runs parlhip_frame_post_u8(flags = obs_flags) into ring slot `new_slot` after
 * this call; an env that completed a step gets cur_slot[e] = new_slot, link[new_slot][e] = the slot
 * of its previous observation (i32 [S,E]; launches an env sat out leave gaps in the ring), since
 * [new_slot][e] = FrameStack's count of valid older frames (u8 [S,E]); row_slot i32 [rows_ring,E] =
 * the slot of the observation a row acted on.  Gather stacks with parlhip_stack_gather_ring_u8.   */
int parlhip_atari_vec_step_elastic(void* states, const uint32_t* rom_table_dev, uint32_t rom_size,
                                   int game, const int64_t* actions, uint8_t* frames, float* rewards,
                                   uint8_t* dones, uint8_t* obs_flags, float* ep_returns,
                                   int32_t* ep_lengths, int E, uint64_t seed, uint64_t env_id0,
                                   int64_t max_episode_steps, const void* reset_cache_dev,
                                   int32_t* jam_flag_dev, int frame_budget, int launch, int rows_limit,
                                   int rows_ring, int batch_rows, int32_t* rows_done,
                                   int32_t* row_launch, int32_t* row_slot, uint8_t* ctl,
                                   int32_t* finished, float* rewards_rows, uint8_t* dones_rows,
                                   int new_slot, int32_t* cur_slot, int32_t* link, uint8_t* since,
                                   parlhip_stream_t stream);
/* The elastic launch with the observation of the envs that completed a step made in the same launch
 * (parlhip_atari_vec_step_obs's tail; obs_out = ring slot `new_slot`, u8 [E, dim*dim]): replaces the
 * parlhip_frame_post_u8(flags = obs_flags) call behind parlhip_atari_vec_step_elastic.  Same outputs.
 * PARLHIP_ENOSUP unless dim is 42 or 84 and the cartridge an unbanked 2K one.                            */
int parlhip_atari_vec_step_elastic_obs(void* states, const uint32_t* rom_table_dev, uint32_t rom_size,
                                       int game, const int64_t* actions, uint8_t* frames, float* rewards,
                                       uint8_t* dones, uint8_t* obs_flags, float* ep_returns,
                                       int32_t* ep_lengths, int E, uint64_t seed, uint64_t env_id0,
                                       int64_t max_episode_steps, const void* reset_cache_dev,
                                       int32_t* jam_flag_dev, int frame_budget, int launch, int rows_limit,
                                       int rows_ring, int batch_rows, int32_t* rows_done,
                                       int32_t* row_launch, int32_t* row_slot, uint8_t* ctl,
                                       int32_t* finished, float* rewards_rows, uint8_t* dones_rows,
                                       int new_slot, int32_t* cur_slot, int32_t* link, uint8_t* since,
                                       uint8_t* obs_out, int dim, const void* tables_dev,
                                       parlhip_stream_t stream);

/* ------------------------------------------------------------------------------------
 * FrameStack (atari_wrappers.py:270-306) over a rollout ring of SINGLE frames
 * ------------------------------------------------------------------------------------ */
/* since_next[e] = (obs_flags[e] & 2) ? 0 : min(since_prev[e] + 1, 3); since_prev NULL = 0.   */
int parlhip_stack_since_update_u8(const uint8_t* obs_flags, const uint8_t* since_prev,
                                  uint8_t* since_next, int E, parlhip_stream_t stream);
/* ring u8 [S,E,frame_bytes], since u8 [S,E]; for sample i: slot = slots[i], env = envs[i];
 * out[i][j] = ring[slot - min(3-j, since[slot][env])][env], j = 0 (oldest) .. 3 (newest).
 * out u8 [n,4,frame_bytes] — the NCHW stack FrameStack._get_ob returns.                      */
int parlhip_stack_gather_u8(const uint8_t* ring, const uint8_t* since, int E, int frame_bytes,
                            const int32_t* slots, const int32_t* envs, int64_t n, uint8_t* out,
                            parlhip_stream_t stream);
/* the same over a CIRCULAR ring of num_slots slots: slot - k wraps below 0; with link != NULL
 * (i32 [S,E], parlhip_atari_vec_step_elastic) the older frames are found by following the env's
 * links instead of stepping one slot back.                                                      */
int parlhip_stack_gather_ring_u8(const uint8_t* ring, const uint8_t* since, const int32_t* link,
                                 int num_slots, int E, int frame_bytes, const int32_t* slots,
                                 const int32_t* envs, int64_t n, uint8_t* out, parlhip_stream_t stream);

/* MonitorEnv.next_episode_results (atari_wrappers.py:88-95) reduced on the device: for the
 * episodes parlhip_atari_vec_step reported closed this step (ep_lengths[e] > 0):
 * acc3[0] += count, acc3[1] += sum of unclipped returns, acc3[2] += sum of lengths (f64).   */
int parlhip_episode_stats_accum_f64(const float* ep_returns, const int32_t* ep_lengths, int E,
                                    double* acc3, parlhip_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Actor-side network trunk: conv1 + conv2 of the IMPALA Atari model on the matrix cores
 * ------------------------------------------------------------------------------------ */
/* examples/IMPALA/atari_model.py:59-71 (AtariModel.policy/value trunk), first two layers:
 * x = obs / 255; conv1 4->16 k4 s2 p1 + ReLU (42x42 -> 21x21); conv2 16->32 k4 s2 p2 + ReLU
 * (-> 11x11).  obs u8 [n,4,42,42] (the stacked observations of the rollout ring), w1 f32
 * [16,4,4,4], b1 [16], w2 f32 [32,16,4,4], b2 [32] (nn.Conv2d / paddle Conv2D layout), out f32
 * [n, 32*11*11] in NCHW flatten order (the input of conv3 = a 3872->256 linear layer).
 * Inference only (no gradient): the actors' forward pass.  One fused MFMA kernel, nothing but
 * obs in / activations out touches HBM.                                                      */
int parlhip_atari42_conv12_u8_f32(const uint8_t* obs, const float* w1, const float* b1,
                                  const float* w2, const float* b2, float* out, int n_obs,
                                  parlhip_stream_t stream);

/* The same two layers for the CURRENT observation of every env of a rollout ring, read in place
 * (parl/env/atari_wrappers.py FrameStack as the ring keeps it, no materialised stack): ring u8
 * [num_slots, E, 42*42] single frames, since u8 [num_slots, E] = steps since the env's last reset,
 * clamped at 3; frame j (0 = oldest) of env n's stack at `slot` is the frame min(3 - j, since[slot][n])
 * slots back (circular) — the rule of parlhip_stack_gather_ring_u8 without link table.  out f32 [E, 3872].
 * Bit-identical to parlhip_stack_gather_ring_u8 + parlhip_atari42_conv12_u8_f32.                       */
int parlhip_atari42_conv12_ring_u8_f32(const uint8_t* ring, const uint8_t* since, int num_slots, int E,
                                       int slot, const float* w1, const float* b1, const float* w2,
                                       const float* b2, float* out, parlhip_stream_t stream);
/* The two weight matrices in the kernel's operand order (what a wavefront keeps in registers: 144 values per
 * lane), so that a workgroup fetches them with 36 coalesced 1 KB loads instead of 144 loads that touch 16 cache
 * lines each — the start-up of a workgroup, half of the actors' 1024-observation launch.  packed_out f32
 * [parlhip_atari42_conv12_weights_bytes() / 4], 16-byte aligned; rebuild whenever w1 / w2 change (the actors:
 * once per weight refresh, atari_model.py:59-71's parameters).  The _packed_ entries below are the two forward
 * entries above with `packed` in place of (w1, w2): bit-identical outputs.  The buffer also carries the backward
 * kernel's operands (parlhip_atari42_conv12_bwd_packed_f32).                                                  */
size_t parlhip_atari42_conv12_weights_bytes(void);
int parlhip_atari42_conv12_weights_f32(const float* w1, const float* w2, float* packed_out,
                                       parlhip_stream_t stream);
int parlhip_atari42_conv12_packed_u8_f32(const uint8_t* obs, const float* packed, const float* b1,
                                         const float* b2, float* out, int n_obs, parlhip_stream_t stream);
int parlhip_atari42_conv12_ring_packed_u8_f32(const uint8_t* ring, const uint8_t* since, int num_slots, int E,
                                              int slot, const float* packed, const float* b1, const float* b2,
                                              float* out, parlhip_stream_t stream);

/* The LEARNER's gradient of the same two layers (IMPALA.learn, parl/algorithms/paddle/impala/
 * impala.py:148-149,205-215 backpropagates through AtariModel.policy / .value; the reference leaves
 * it to the framework's conv backward): d loss / d (w1, b1, w2, b2) given the forward output
 * a2 = relu(conv2(relu(conv1(obs/255)))) [n,3872] and dy = d loss / d a2 [n,3872].  conv1 is
 * recomputed on the fly (bit-identical to the forward kernel), so no activation other than a2 is
 * stored; per-workgroup partial sums go to `workspace` (parlhip_atari42_conv12_bwd_workspace_bytes)
 * and are added in a fixed order: the result is deterministic.  Outputs are OVERWRITTEN:
 * dw1 [16,64], db1 [16], dw2 [32,256], db2 [32].  The observations get no gradient.           */
size_t parlhip_atari42_conv12_bwd_workspace_bytes(int n_obs);
int parlhip_atari42_conv12_bwd_f32(const uint8_t* obs, const float* w1, const float* b1,
                                   const float* w2, const float* a2, const float* dy, int n_obs,
                                   float* workspace, float* dw1, float* db1, float* dw2, float* db2,
                                   parlhip_stream_t stream);
/* The same with the weights from parlhip_atari42_conv12_weights_f32's buffer (of the CURRENT weights) in place of
 * (w1, w2): bit-identical gradients, 12 coalesced operand loads per lane instead of 48 scattered ones.        */
int parlhip_atari42_conv12_bwd_packed_f32(const uint8_t* obs, const float* packed, const float* b1,
                                          const float* a2, const float* dy, int n_obs, float* workspace,
                                          float* dw1, float* db1, float* dw2, float* db2,
                                          parlhip_stream_t stream);

/* The learner's pair with the conv1 activation SAVED instead of recomputed (round 6; same reference lines: the
 * framework's autograd keeps relu(conv1) of atari_model.py:59-71 for the backward pass — so does this pair).  The
 * forward writes, next to `out`, the zero-padded conv1 tile of every observation as it stands in LDS
 * (a1_out f32, parlhip_atari42_conv12_a1_bytes(n_obs) bytes = 16 x 25 x 25 floats per observation, 16-byte aligned;
 * `out` bit-identical to the entries above); the backward reads it back with an LDS-DMA copy in place of its
 * conv1 recompute (448 of its 2,812 MFMAs per observation and the slowest of its phases) and runs its dz1 / dW1
 * sums as two interleaved accumulator chains — gradients equal to parlhip_atari42_conv12_bwd_packed_f32's up to the
 * order of those sums (deterministic, run-to-run bit-identical).  Meant for the learner's 1000-row updates
 * (40 MB per update); at 40 KB per observation a 51,200-row pass would move 2 GB each way: use the recompute there. */
size_t parlhip_atari42_conv12_a1_bytes(int n_obs);
int parlhip_atari42_conv12_packed_save_u8_f32(const uint8_t* obs, const float* packed, const float* b1,
                                              const float* b2, float* out, float* a1_out, int n_obs,
                                              parlhip_stream_t stream);
int parlhip_atari42_conv12_bwd_saved_f32(const uint8_t* obs, const float* packed, const float* b1, const float* a1,
                                         const float* a2, const float* dy, int n_obs, float* workspace,
                                         float* dw1, float* db1, float* dw2, float* db2, parlhip_stream_t stream);

/* examples/A2C/atari_model.py:21-104 (AtariModel trunk), first layer — the 84x84 -> 20x20
 * contraction: x = obs / 255; conv1 4->32 k8 s4 p1 + ReLU.  obs u8 [n,4,84,84], w1 f32
 * [32,4,8,8], b1 [32] (nn.Conv2d layout), out f32 [n,32,20,20] (NCHW, the input of conv2).
 * Inference only (the actors' / bootstrap-value forward).  Implicit GEMM [400 x 256] x [256 x 32]
 * per observation on v_mfma_f32_16x16x4_f32; obs must be 4-byte and out 16-byte aligned.       */
int parlhip_atari84_conv1_u8_f32(const uint8_t* obs, const float* w1, const float* b1, float* out,
                                 int n_obs, parlhip_stream_t stream);
/* The same layer for the CURRENT observation of every env of an 84x84 rollout ring, read in place (ring u8
 * [num_slots, E, 84*84], since u8 [num_slots, E]; the frame rule of parlhip_atari42_conv12_ring_u8_f32).
 * out f32 [E,32,20,20].  Bit-identical to parlhip_stack_gather_ring_u8 + parlhip_atari84_conv1_u8_f32.        */
int parlhip_atari84_conv1_ring_u8_f32(const uint8_t* ring, const uint8_t* since, int num_slots, int E, int slot,
                                      const float* w1, const float* b1, float* out, parlhip_stream_t stream);
/* The two entries above with the weight matrix in operand order: wt1 f32 [64][2][64], wt1[ks][nt][q*16 + col] =
 * w1[16*nt + col][4*ks + q] over w1 viewed as [32, 256] (the layout parlhip_atari84_conv23_f32 takes its wt2 / wt3
 * in): a workgroup fetches its operands with coalesced loads — the scattered fetch from the nn.Conv2d layout is
 * the start-up cost of a workgroup (see parlhip_atari42_conv12_weights_f32).  Bit-identical outputs.          */
int parlhip_atari84_conv1_packed_u8_f32(const uint8_t* obs, const float* wt1, const float* b1, float* out,
                                        int n_obs, parlhip_stream_t stream);
int parlhip_atari84_conv1_ring_packed_u8_f32(const uint8_t* ring, const uint8_t* since, int num_slots, int E,
                                             int slot, const float* wt1, const float* b1, float* out,
                                             parlhip_stream_t stream);

/* examples/A2C/atari_model.py:21-104 (AtariModel trunk), second and third layer, fused:
 * conv2 32->64 k4 s2 p2 + ReLU (20x20 -> 11x11), conv3 64->64 k3 s1 + ReLU (-> 9x9).
 * a1 f32 [n,32,20,20] (the output of parlhip_atari84_conv1_u8_f32, 16-byte aligned), out a3 f32
 * [n, 64*9*9] in NCHW flatten order (the input of the 5184->512 fc layer); a2_out (optional, may be
 * NULL) receives the conv2 activation [n,64,11,11] the backward pass needs.  The weights arrive in
 * MFMA operand order, streamed from L2 by the kernel:
 *   wt2[ks][nt][lane] = w2[16 nt + (lane & 15)][4 ks + (lane >> 4)]          ks < 128, nt < 4
 *   wt3[ks][nt][lane] = w3[16 nt + (lane & 15)][c][kh][kw],  4 ks + (lane >> 4) = (3 kh + kw) 64 + c
 * (w2 = conv2.weight.flatten(1), nn.Conv2d layout).                                              */
int parlhip_atari84_conv23_f32(const float* a1, const float* wt2, const float* b2, const float* wt3,
                               const float* b3, float* a2_out, float* a3_out, int n_obs,
                               parlhip_stream_t stream);

/* Backward of conv3 (64->64 k3 s1) of the same network for the learner: given the saved activations
 * a2 [n,64,11,11] and a3 [n,5184] and dy3 = d loss / d a3 [n,5184]:
 *   dz2 [n,64,11,11] = (d loss / d a2) * (a2 > 0)   (the input of parlhip_atari84_conv2_bwd_f32)
 *   dw3_db3 [64*576 + 64]: d loss / d w3 as [o][k'] with k' = (3 kh + kw) 64 + c, then d loss / d b3
 * wt3b = the B operand of the transposed convolution in MFMA order:
 *   wt3b[ks][nt][lane] = w3[o][16 nt + (lane & 15)][kh][kw],  4 ks + (lane >> 4) = (3 kh + kw) 64 + o
 * workspace: parlhip_atari84_conv3_bwd_workspace_bytes(n) bytes; deterministic.                   */
size_t parlhip_atari84_conv3_bwd_workspace_bytes(int n_obs);
int parlhip_atari84_conv3_bwd_f32(const float* a2, const float* a3, const float* dy3, const float* wt3b,
                                  int n_obs, float* workspace, float* dz2, float* dw3_db3,
                                  parlhip_stream_t stream);

/* Backward of conv2 (32->64 k4 s2 p2): a1 [n,32,20,20] saved by the forward, dz2 from the conv3
 * backward -> dz1 [n,32,20,20] = (d loss / d a1) * (a1 > 0), dw2_db2 [64*512 + 64] (d w2 as
 * [64][c*16 + kh*4 + kw], then d b2).  wt2b[cls][o][nt][lane] = w2[o][16 nt + (lane & 15)]
 * [py + 2 (lane >> 5)][px + 2 ((lane >> 4) & 1)] for the output parity class cls = 2 py + px.          */
size_t parlhip_atari84_conv2_bwd_workspace_bytes(int n_obs);
int parlhip_atari84_conv2_bwd_f32(const float* a1, const float* dz2, const float* wt2b, int n_obs,
                                  float* workspace, float* dz1, float* dw2_db2, parlhip_stream_t stream);
/* Backward of conv1 (4->32 k8 s4 p1) w.r.t. its parameters: obs u8 [n,4,84,84], dz1 from the conv2
 * backward -> dw1_db1 [32*256 + 32] (d w1 as [32][ci*64 + kh*8 + kw], then d b1).                  */
size_t parlhip_atari84_conv1_bwd_workspace_bytes(int n_obs);
int parlhip_atari84_conv1_bwd_f32(const uint8_t* obs, const float* dz1, int n_obs, float* workspace,
                                  float* dw1_db1, parlhip_stream_t stream);

/* ------------------------------------------------------------------------------------
 * PPO: running observation / return normalisation and the minibatch gather
 * ------------------------------------------------------------------------------------ */
/* VecNormalizeEnv._obfilt (parl/env/mujoco_wrappers.py:140-156) for E host-stepped envs at once,
 * each with its OWN RunningMeanStd (mujoco_wrappers.py:73-92; one VecNormalizeEnv per env,
 * examples/PPO/env_utils.py:118-127), i.e. update_mean_var_count_from_moments
 * (mujoco_wrappers.py:185-206) with batch_mean = x, batch_var = 0, batch_count = 1, in float64
 * with numpy's operation order (bit-identical statistics and outputs).
 * raw f64 [E,D]; mean, var f64 [E,D] and count f64 [E] are updated in place when update != 0
 * (training mode); mask u8 [E] or NULL selects the envs to process (the reset path,
 * env_utils.py:95-103: only envs that just finished filter their reset observation);
 * out f32 [E,D] = the cast RolloutStorage.append performs (examples/PPO/storage.py:36) and / or
 * out64 f64 [E,D]; rows of unselected envs are left untouched.                                   */
int parlhip_vecnorm_obs_f64(const double* raw, double* mean, double* var, double* count,
                            const uint8_t* mask, float* out, double* out64, int E, int D,
                            double clipob, double eps, int update, parlhip_stream_t stream);
/* VecNormalizeEnv.step, reward half (mujoco_wrappers.py:120-136): ret = ret*gamma + rew;
 * ret_rms.update(ret); rew = clip(rew / sqrt(ret_rms.var + eps), -cliprew, cliprew);
 * ret = 0 where done.  rew f64 [E], done u8 [E]; ret, ret_mean, ret_var, ret_count f64 [E] in
 * place; out f32 [E] and / or out64 f64 [E].                                                     */
int parlhip_vecnorm_reward_f64(const double* rew, const uint8_t* done, double* ret,
                               double* ret_mean, double* ret_var, double* ret_count, float* out,
                               double* out64, int E, double gamma, double cliprew, double eps,
                               parlhip_stream_t stream);
/* RolloutStorage.sample_batch (examples/PPO/storage.py:66-76) for one minibatch index
 * (examples/PPO/agent.py:91-99): out_x[m] = x[idx[m]] for the six flattened rollout arrays in one
 * launch.  obs f32 [N,obs_dim], actions f32 [N,act_dim] (act_dim 0: scalar actions are passed as
 * act_dim 1), the others f32 [N]; idx i64 [M].  An index outside [0,N) raises the device data-
 * error flag (parlhip_consume_device_errors) and writes nothing for that row.                    */
int parlhip_ppo_sample_batch_f32(const float* obs, const float* actions, const float* logprobs,
                                 const float* advantages, const float* returns, const float* values,
                                 const int64_t* idx, float* out_obs, float* out_actions,
                                 float* out_logprobs, float* out_advantages, float* out_returns,
                                 float* out_values, int64_t N, int64_t M, int obs_dim, int act_dim,
                                 parlhip_stream_t stream);

/* ------------------------------------------------------------------------------------
 * The learner's parameter update: global-norm gradient clipping + Adam
 * parl/algorithms/paddle/impala.py:113-117 (Adam(learning_rate, grad_clip=ClipGradByGlobalNorm(40))),
 * parl/algorithms/torch/a2c.py:76-78 (clip_grad_norm_(parameters, 40); optimizer.step())
 * ------------------------------------------------------------------------------------ */
/* n_tensors <= 16 parameter tensors (more: PARLHIP_ENOSUP, callers keep the framework's optimizer).  HOST arrays of
 * n_tensors DEVICE pointers: params, grads, exp_avg, exp_avg_sq float32 [numel[i]], steps float32 [1] each (torch's
 * capturable Adam state: the step counter of a parameter is a float32 scalar tensor); lr: DEVICE float32 scalar.
 * Two launches on `stream`, no host synchronisation (capturable in a hipGraph: pointers and scalars are kernel
 * arguments): steps[i] += 1; norm = sqrt(sum of g^2 over all tensors); g *= min(1, max_norm / (norm + 1e-6)) in
 * place (torch.nn.utils.clip_grad_norm_); m += (1-beta1)(g-m); v = beta2 v + (1-beta2) g^2;
 * p -= lr / (1-beta1^step) * m / (sqrt(v) / sqrt(1-beta2^step) + eps) (torch.optim.Adam, no weight decay / amsgrad;
 * the hyper-parameters are doubles as torch's are: 1 - beta and the bias corrections are formed in double, the element
 * arithmetic is float32).
 * workspace: parlhip_clip_adam_workspace_bytes(...) bytes; norm_out: optional DEVICE float32 [1] (the norm before
 * clipping).  Deterministic (per-workgroup partial sums added in one fixed order).                                */
size_t parlhip_clip_adam_workspace_bytes(int n_tensors, const int64_t* numel);
int parlhip_clip_adam_f32(int n_tensors, float* const* params, float* const* grads, float* const* exp_avg,
                          float* const* exp_avg_sq, float* const* steps, const int64_t* numel, const float* lr,
                          double beta1, double beta2, double eps, double max_norm, float* workspace, float* norm_out,
                          parlhip_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PARL_HIP_H_ */
