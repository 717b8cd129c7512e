/*
 * atari_env_oracle.c — CPU restatement of the reference's per-env wrapper chain and VectorEnv
 * auto-reset, on top of the oracle emulator.  TEST INFRASTRUCTURE ONLY.
 *
 * Follows (paths relative to the PARL tree), innermost first — wrap_deepmind order,
 * parl/env/atari_wrappers.py:356-385:
 *   gym TimeLimit (gym 0.12.1, max_episode_steps = 400000 for *NoFrameskip-v4)   [third party]
 *   CompatWrapper.step        parl/env/compat_wrappers.py:85-99   (never-reset step counter)
 *   MonitorEnv                parl/env/atari_wrappers.py:44-100
 *   NoopResetEnv.reset        :114-130
 *   MaxAndSkipEnv.step        :223-240
 *   EpisodicLifeEnv           :177-211
 *   FireResetEnv.reset        :163-171
 *   WarpFrame                 :246-267  (oracle/frame_oracle.c)
 *   ClipRewardEnv.reward      :149-151
 *   FrameStack                :270-306  (NCHW, oldest first)
 *   VectorEnv.step            parl/env/vector_env.py:41-63  (auto-reset, returns reset obs)
 *
 * Randomness: the reference seeds nothing; here the noop count of reset number n of
 * env e is 1 + philox4x32-10(seed; n, e)[0] % 30, the same stream the device uses.
 * Known simplification (both here and on the device): a CompatWrapper /
 * TimeLimit step-limit `done` that fires INSIDE a reset sequence is handled exactly as the
 * wrappers do, because the sequence is executed step by step here; the device must match.
 */
#include "atari_oracle.h"
#include <stdlib.h>
#include <string.h>

void oracle_philox4x32_10(uint64_t seed, uint64_t offset, uint64_t row, uint32_t out[4]);
int oracle_frame_post_u8(const uint8_t* f0, const uint8_t* f1, int fmt, uint8_t* out,
                         int64_t out_stride, int E, int dim, const void* blob);
size_t oracle_frame_tables_bytes(int dim);
int oracle_frame_tables_init(void* blob, int dim);

#define MAX_EP_QUEUE 64

typedef struct {
  Ale ale;
  int n_actions, action_set[18], has_fire;
  uint8_t obs_buf[2][ATARI_FRAME_BYTES]; /* MaxAndSkipEnv._obs_buffer */
  uint8_t last_frame[ATARI_FRAME_BYTES]; /* raw obs of the last emulated frame */
  int obs_is_single;                     /* reset obs came straight from NoopResetEnv */
  int lives, was_real_done;              /* EpisodicLifeEnv */
  int has_episode;                       /* MonitorEnv: _current_reward is not None */
  double cur_reward;
  int64_t num_steps;
  int64_t total_steps;                   /* MonitorEnv._total_steps: every raw step of the run (:77) */
  int64_t elapsed_steps, compat_count, max_episode_steps;
  int skip;
  uint64_t seed, env_id, reset_count;
  int dim;
  uint8_t* stack; /* FrameStack: 4 frames of dim*dim, oldest first */
  double ep_rewards[MAX_EP_QUEUE];
  int64_t ep_lengths[MAX_EP_QUEUE];
  int ep_n;
} EnvO;

typedef struct {
  int E, dim;
  EnvO* envs;
  uint8_t* rom;
  void* tables;
} VecO;

/* ---- innermost: TimeLimit(AtariEnv) + CompatWrapper + MonitorEnv ---- */
static int raw_step(EnvO* v, int ale_action, int* reward) {
  int r = ale_act(&v->ale, ale_action, v->last_frame);
  int done = v->ale.terminal;
  v->elapsed_steps++;                                   /* gym TimeLimit.step */
  if (v->elapsed_steps >= v->max_episode_steps) done = 1;
  v->compat_count++;                                    /* compat_wrappers.py:86,96-98 */
  if (v->compat_count >= v->max_episode_steps) { done = 1; v->compat_count = 0; }
  v->cur_reward += r;                                   /* atari_wrappers.py:75-77 */
  v->num_steps++;
  v->total_steps++;
  *reward = r;
  return done;
}

static void monitor_reset(EnvO* v) { /* MonitorEnv.reset :57-71 -> ... -> ALE reset_game */
  ale_reset(&v->ale, v->last_frame);
  v->elapsed_steps = 0;
  if (v->has_episode && v->ep_n < MAX_EP_QUEUE) {
    v->ep_rewards[v->ep_n] = v->cur_reward;
    v->ep_lengths[v->ep_n] = v->num_steps;
    v->ep_n++;
  }
  v->has_episode = 1;
  v->cur_reward = 0;
  v->num_steps = 0;
}

static void noop_reset(EnvO* v) { /* NoopResetEnv.reset :114-130 */
  monitor_reset(v);
  uint32_t w[4];
  oracle_philox4x32_10(v->seed, v->reset_count, v->env_id, w);
  v->reset_count++;
  const int noops = 1 + (int)(w[0] % 30u);
  for (int i = 0; i < noops; ++i) {
    int r;
    if (raw_step(v, ACT_NOOP, &r)) monitor_reset(v);
  }
  v->obs_is_single = 1;
}

static int maxskip_step(EnvO* v, int ale_action, double* total) { /* MaxAndSkipEnv.step :223-240 */
  int done = 0;
  *total = 0.0;
  for (int i = 0; i < v->skip; ++i) {
    int r;
    done = raw_step(v, ale_action, &r);
    if (i == v->skip - 2) memcpy(v->obs_buf[0], v->last_frame, ATARI_FRAME_BYTES);
    if (i == v->skip - 1) memcpy(v->obs_buf[1], v->last_frame, ATARI_FRAME_BYTES);
    *total += r;
    if (done) break;
  }
  v->obs_is_single = 0;
  return done;
}

static int episodic_step(EnvO* v, int ale_action, double* reward) { /* EpisodicLifeEnv.step :186-198 */
  int done = maxskip_step(v, ale_action, reward);
  v->was_real_done = done;
  const int lives = v->ale.lives;
  if (lives < v->lives && lives > 0) done = 1;
  v->lives = lives;
  return done;
}

static void episodic_reset(EnvO* v) { /* EpisodicLifeEnv.reset :200-211 */
  if (v->was_real_done) {
    noop_reset(v);
  } else {
    double r;
    maxskip_step(v, ACT_NOOP, &r);
  }
  v->lives = v->ale.lives;
}

static void fire_reset(EnvO* v) { /* FireResetEnv.reset :163-171 (or passthrough) */
  episodic_reset(v);
  if (!v->has_fire) return;
  double r;
  if (episodic_step(v, v->action_set[1], &r)) episodic_reset(v);
  /* the obs returned is the one of step(2) even when a reset follows it (:168-171) */
  int done = episodic_step(v, v->action_set[2], &r);
  if (done) {
    /* keep step(2)'s observation: snapshot the buffers before the reset overwrites them */
    uint8_t* keep = (uint8_t*)malloc(2 * ATARI_FRAME_BYTES);
    memcpy(keep, v->obs_buf, 2 * ATARI_FRAME_BYTES);
    episodic_reset(v);
    memcpy(v->obs_buf, keep, 2 * ATARI_FRAME_BYTES);
    v->obs_is_single = 0;
    free(keep);
  }
}

static void warp_current(VecO* V, EnvO* v, uint8_t* out) { /* max (:239) + WarpFrame (:263-267) */
  if (v->obs_is_single)
    oracle_frame_post_u8(v->last_frame, 0, 1, out, 0, 1, V->dim, V->tables);
  else
    oracle_frame_post_u8(v->obs_buf[0], v->obs_buf[1], 1, out, 0, 1, V->dim, V->tables);
}

static void stack_reset(VecO* V, EnvO* v) { /* FrameStack.reset :290-294 */
  const size_t n = (size_t)V->dim * V->dim;
  fire_reset(v);
  warp_current(V, v, v->stack);
  for (int k = 1; k < 4; ++k) memcpy(v->stack + k * n, v->stack, n);
}

/* ---- public API (ctypes) ---- */
void* oracle_vec_new(const uint8_t* rom, uint32_t rom_size, int game, int E, int dim,
                     uint64_t seed, uint64_t env_id0, int64_t max_episode_steps) {
  VecO* V = (VecO*)calloc(1, sizeof(VecO));
  V->E = E; V->dim = dim;
  V->rom = (uint8_t*)malloc(rom_size);
  memcpy(V->rom, rom, rom_size);
  V->tables = malloc(oracle_frame_tables_bytes(dim));
  oracle_frame_tables_init(V->tables, dim);
  V->envs = (EnvO*)calloc((size_t)E, sizeof(EnvO));
  for (int e = 0; e < E; ++e) {
    EnvO* v = &V->envs[e];
    ale_init(&v->ale, V->rom, rom_size, game);
    v->n_actions = ale_minimal_actions(game, v->action_set);
    v->has_fire = v->n_actions >= 3 && v->action_set[1] == ACT_FIRE;
    v->was_real_done = 1;
    v->skip = 4;
    v->seed = seed; v->env_id = env_id0 + (uint64_t)e;
    v->max_episode_steps = max_episode_steps;
    v->dim = dim;
    v->stack = (uint8_t*)calloc(4, (size_t)dim * dim);
  }
  return V;
}

void oracle_vec_free(void* p) {
  VecO* V = (VecO*)p;
  for (int e = 0; e < V->E; ++e) free(V->envs[e].stack);
  free(V->envs); free(V->rom); free(V->tables); free(V);
}

int oracle_vec_num_actions(void* p) { return ((VecO*)p)->envs[0].n_actions; }

/* VectorEnv.reset vector_env.py:34-39 -> obs u8 [E,4,dim,dim] */
void oracle_vec_reset(void* p, uint8_t* obs) {
  VecO* V = (VecO*)p;
  const size_t n = (size_t)V->dim * V->dim * 4;
  for (int e = 0; e < V->E; ++e) {
    stack_reset(V, &V->envs[e]);
    memcpy(obs + e * n, V->envs[e].stack, n);
  }
}

/* VectorEnv.step vector_env.py:41-63 */
void oracle_vec_step(void* p, const int64_t* actions, uint8_t* obs, float* rewards,
                     uint8_t* dones) {
  VecO* V = (VecO*)p;
  const size_t n1 = (size_t)V->dim * V->dim, n = n1 * 4;
  for (int e = 0; e < V->E; ++e) {
    EnvO* v = &V->envs[e];
    double r;
    int a = (int)actions[e];
    if (a < 0 || a >= v->n_actions) a = 0;
    int done = episodic_step(v, v->action_set[a], &r);
    /* WarpFrame + FrameStack.step :296-299 */
    memmove(v->stack, v->stack + n1, 3 * n1);
    warp_current(V, v, v->stack + 3 * n1);
    rewards[e] = (float)((r > 0) - (r < 0)); /* ClipRewardEnv: np.sign */
    dones[e] = (uint8_t)done;
    if (done) stack_reset(V, v);             /* VectorEnv auto-reset :56-57 */
    memcpy(obs + e * n, v->stack, n);
  }
}

/* MonitorEnv.next_episode_results via Actor.get_metrics (examples/IMPALA/actor.py:93-101) */
int oracle_vec_pop_episodes(void* p, int env, double* rewards, int64_t* lengths, int cap) {
  VecO* V = (VecO*)p;
  EnvO* v = &V->envs[env];
  int n = v->ep_n < cap ? v->ep_n : cap;
  memcpy(rewards, v->ep_rewards, sizeof(double) * (size_t)n);
  memcpy(lengths, v->ep_lengths, sizeof(int64_t) * (size_t)n);
  v->ep_n = 0;
  return n;
}

/* introspection for parity tests */
void oracle_vec_ram(void* p, int env, uint8_t* out) { memcpy(out, ((VecO*)p)->envs[env].ale.emu.ram, 128); }
void oracle_vec_raw_frames(void* p, int env, uint8_t* out) {
  memcpy(out, ((VecO*)p)->envs[env].obs_buf, 2 * ATARI_FRAME_BYTES);
}
int oracle_vec_lives(void* p, int env) { return ((VecO*)p)->envs[env].lives; }
/* MonitorEnv.get_total_steps atari_wrappers.py:73-77,90-91 */
int64_t oracle_vec_total_steps(void* p, int env) { return ((VecO*)p)->envs[env].total_steps; }
