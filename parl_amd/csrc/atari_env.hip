// atari_env.hip — vectorised Atari env step for gfx950: ALE layer + the reference's wrapper
// chain + VectorEnv auto-reset around the one-env-per-wavefront emulator (atari_core.hpp).
//
// Reference call sites this replaces (paths relative to the PARL tree):
//   parl/env/vector_env.py:41-63            VectorEnv.step (auto-reset, returns the reset obs)
//   parl/env/atari_wrappers.py:356-385      wrap_deepmind order; :223-240 MaxAndSkipEnv;
//                                           :177-211 EpisodicLifeEnv; :114-130 NoopResetEnv;
//                                           :163-171 FireResetEnv; :44-100 MonitorEnv;
//                                           :149-151 ClipRewardEnv
//   parl/env/compat_wrappers.py:85-99       CompatWrapper step counter
//   examples/IMPALA/actor.py:66-67,95-101   the caller (vector_env.step, get_metrics)
// Third party behind them (ALE via atari-py, gym TimeLimit): restated, see oracle/atari_oracle.h.
#include "common.hpp"
#include "atari_core.hpp"
#include "philox.hpp"
#include <math.h>
#include <string.h>
#include <vector>

namespace parlhip {
namespace atari {

constexpr int kNumSnap = 30;  // NoopResetEnv noop_max
constexpr size_t kSnapBytes = kStateBytes + 2 * (size_t)kFrameBytes;  // 67,712 (16-B multiple)
static_assert(kSnapBytes % 16 == 0, "snapshot stride must keep 16-byte alignment");

struct EnvParams {
  int game, rom_size, E;
  unsigned long long seed, env_id0;
  long long max_episode_steps;
};

// wrapper-level + ALE-level state that rides along the emulator (all wave-uniform)
struct Env {
  Emu emu;
  int game;
  // ALE
  int paddle, score, terminal, ale_lives, started, frame_number;
  // wrappers
  int lives, was_real_done, has_episode, cur_reward, num_steps, reset_count, obs_single;
  long long elapsed, compat_count, max_steps;
  unsigned long long seed, env_id;
  // per-call outputs
  int ep_closed, ep_return, ep_length;
  int no_render;  // keep the obs buffers untouched (reset that follows FireResetEnv's step(2))
  uint8_t* buf0;
  uint8_t* buf1;

  DEVI int action_code(int idx) const {  // ALE minimal action sets (Pong 6, Breakout 4)
    switch (idx) {
      case 0: return ACT_NOOP;
      case 1: return ACT_FIRE;
      case 2: return ACT_RIGHT;
      case 3: return ACT_LEFT;
      case 4: return ACT_RIGHTFIRE;
      default: return ACT_LEFTFIRE;
    }
  }
  DEVI int num_actions() const { return game == GAME_BREAKOUT ? 4 : 6; }

  DEVI void apply_action(int act) {  // ALEState::applyActionPaddles
    int delta = 0, fire = 0;
    emu.sw_reset = act == ACT_RESET;
    if (act == ACT_RIGHT || act == ACT_RIGHTFIRE) delta = -kPaddleDelta;
    if (act == ACT_LEFT || act == ACT_LEFTFIRE) delta = kPaddleDelta;
    if (act == ACT_FIRE || act == ACT_RIGHTFIRE || act == ACT_LEFTFIRE) fire = 1;
    paddle += delta;
    paddle = paddle < kPaddleMin ? kPaddleMin : (paddle > kPaddleMax ? kPaddleMax : paddle);
    const bool swap = game == GAME_PONG;  // Stella props: Video Olympics SwapPaddles=YES
    emu.paddle_res0 = swap ? kPaddleDefault : paddle;
    emu.paddle_res1 = swap ? paddle : kPaddleDefault;
    emu.fire0 = swap ? 0 : fire;
    emu.fire1 = swap ? fire : 0;
  }

  DEVI int rom_step() {  // RomSettings::step (Pong.cpp / Breakout.cpp)
    int reward = 0;
    if (game == GAME_PONG) {
      const int x = emu.ram_rd(13), y = emu.ram_rd(14);
      const int sc = y - x;
      reward = sc - score;
      score = sc;
      terminal = (x == 21 || y == 21);
      ale_lives = 0;
    } else if (game == GAME_BREAKOUT) {
      const int x = emu.ram_rd(77), y = emu.ram_rd(76);
      const int sc = (x & 0x0f) + 10 * ((x & 0xf0) >> 4) + 100 * (y & 0x0f);
      reward = sc - score;
      score = sc;
      const int lv = emu.ram_rd(57);
      if (!started && lv == 5) started = 1;
      terminal = started && lv == 0;
      ale_lives = lv;
    }
    return reward;
  }

  DEVI void ale_reset() {  // ALE reset_game(): system reset, 60 NOOP frames, 4 RESET frames
    paddle = kPaddleDefault;
    const uint32_t* romw = emu.romw;
    const int mask = emu.rom_mask, lane = emu.lane;
    emu.system_reset();
    emu.romw = romw; emu.rom_mask = mask; emu.lane = lane;
    for (int i = 0; i < 60; ++i) { apply_action(ACT_NOOP); emu.frame(nullptr); }
    for (int i = 0; i < 4; ++i) { apply_action(ACT_RESET); emu.frame(nullptr); }
    score = 0; terminal = 0; started = 0;
    ale_lives = game == GAME_BREAKOUT ? 5 : 0;
    frame_number = 0;
  }

  // TimeLimit(AtariEnv).step + CompatWrapper.step + MonitorEnv.step
  DEVI bool raw_step(int act, uint8_t* fbp, int& reward) {
    apply_action(act);
    emu.frame(fbp);
    reward = rom_step();
    frame_number++;
    bool done = terminal != 0;
    elapsed++;
    if (elapsed >= max_steps) done = true;
    compat_count++;
    if (compat_count >= max_steps) { done = true; compat_count = 0; }
    cur_reward += reward;
    num_steps++;
    return done;
  }

  DEVI void monitor_reset() {
    ale_reset();
    elapsed = 0;
    if (has_episode) { ep_closed++; ep_return = cur_reward; ep_length = num_steps; }
    has_episode = 1;
    cur_reward = 0;
    num_steps = 0;
  }

  DEVI int draw_noops() {
    uint32_t w[4];
    philox4x32_10(seed, (unsigned long long)(unsigned)reset_count, env_id, w);
    reset_count++;
    return 1 + (int)((uint32_t)rfl((int)w[0]) % 30u);
  }

  DEVI void noop_reset(int noops, bool has_fire) {
    monitor_reset();
    for (int i = 0; i < noops; ++i) {
      int r;
      // the raw obs of the last noop is the reset obs only when no FireResetEnv follows
      uint8_t* fbp = (!has_fire && i == noops - 1 && !no_render) ? buf0 : nullptr;
      if (raw_step(ACT_NOOP, fbp, r)) monitor_reset();
    }
    obs_single = 1;
  }

  DEVI bool maxskip_step(int act, int& total) {
    bool done = false;
    total = 0;
    for (int i = 0; i < 4; ++i) {
      int r;
      uint8_t* fbp = no_render ? nullptr : (i == 2 ? buf0 : (i == 3 ? buf1 : nullptr));
      done = raw_step(act, fbp, r);
      total += r;
      if (done) break;
    }
    obs_single = 0;
    return done;
  }

  DEVI bool episodic_step(int act, int& reward) {
    bool done = maxskip_step(act, reward);
    was_real_done = done;
    if (ale_lives < lives && ale_lives > 0) done = true;
    lives = ale_lives;
    return done;
  }

  DEVI void episodic_reset(int noops, bool has_fire) {
    if (was_real_done) {
      noop_reset(noops, has_fire);
    } else {
      int r;
      maxskip_step(ACT_NOOP, r);
    }
    lives = ale_lives;
  }

  // FireResetEnv.reset + (caller) WarpFrame/FrameStack.reset; noops drawn by the caller once per
  // NoopResetEnv.reset call — a second real reset inside the sequence draws again.
  DEVI void fire_reset(bool has_fire) {
    episodic_reset(was_real_done ? draw_noops() : 0, has_fire);
    if (!has_fire) return;
    int r;
    if (episodic_step(ACT_FIRE, r)) episodic_reset(was_real_done ? draw_noops() : 0, has_fire);
    if (episodic_step(action_code(2), r)) {
      // the reference returns step(2)'s obs and still resets (atari_wrappers.py:168-171):
      // run that reset without touching the obs buffers
      no_render = 1;
      episodic_reset(was_real_done ? draw_noops() : 0, has_fire);
      no_render = 0;
      obs_single = 0;
    }
  }
};

// ---- state blob <-> registers ----
DEVI void load_env(Env& v, const uint8_t* blob, int lane) {
  const int* s = (const int*)(blob + kOffScalars);
  Emu& e = v.emu;
  e.ram_lo = blob[kOffRam + lane];
  e.ram_hi = blob[kOffRam + 64 + lane];
  e.tia = blob[kOffTia + lane];
  auto L = [&](int i) { return rfl(s[i]); };
  e.A = L(S_A); e.X = L(S_X); e.Y = L(S_Y); e.S = L(S_S); e.P = L(S_P); e.PC = L(S_PC);
  e.cyc = L(S_CYC); e.cyc0 = L(S_CYC0); e.last_clock = L(S_LAST_CLOCK);
  e.vsync_finish = L(S_VSYNC_FINISH); e.dump_dis_cyc = L(S_DUMP_DIS_CYC); e.dump_en = L(S_DUMP_EN);
  e.timer = L(S_TIMER); e.timer_shift = L(S_TIMER_SHIFT); e.timer_set_cyc = L(S_TIMER_SET_CYC);
  e.ddra = L(S_DDRA); e.ddrb = L(S_DDRB); e.swcha_out = L(S_SWCHA_OUT); e.swchb_out = L(S_SWCHB_OUT);
  e.cx = L(S_CX); e.jam = L(S_JAM); e.stop = 0;
  e.paddle_res0 = e.paddle_res1 = kPaddleDefault; e.fire0 = e.fire1 = e.sw_reset = 0;
  e.fb = nullptr;
  v.paddle = L(S_PADDLE); v.score = L(S_SCORE); v.terminal = L(S_TERMINAL);
  v.ale_lives = L(S_ALE_LIVES); v.started = L(S_STARTED); v.frame_number = L(S_FRAME_NUMBER);
  v.lives = L(S_LIVES); v.was_real_done = L(S_WAS_REAL_DONE); v.has_episode = L(S_HAS_EPISODE);
  v.cur_reward = L(S_CUR_REWARD); v.num_steps = L(S_NUM_STEPS); v.elapsed = L(S_ELAPSED);
  v.compat_count = L(S_COMPAT_COUNT); v.reset_count = L(S_RESET_COUNT);
  v.obs_single = L(S_OBS_SINGLE);
}

DEVI void store_env(const Env& v, uint8_t* blob, int lane) {
  int* s = (int*)(blob + kOffScalars);
  const Emu& e = v.emu;
  blob[kOffRam + lane] = (uint8_t)e.ram_lo;
  blob[kOffRam + 64 + lane] = (uint8_t)e.ram_hi;
  blob[kOffTia + lane] = (uint8_t)e.tia;
  if (lane == 0) {
    s[S_A] = e.A; s[S_X] = e.X; s[S_Y] = e.Y; s[S_S] = e.S; s[S_P] = e.P; s[S_PC] = e.PC;
    s[S_BUS] = 0;
    s[S_CYC] = e.cyc; s[S_CYC0] = e.cyc0; s[S_LAST_CLOCK] = e.last_clock;
    s[S_VSYNC_FINISH] = e.vsync_finish; s[S_DUMP_DIS_CYC] = e.dump_dis_cyc; s[S_DUMP_EN] = e.dump_en;
    s[S_TIMER] = e.timer; s[S_TIMER_SHIFT] = e.timer_shift; s[S_TIMER_SET_CYC] = e.timer_set_cyc;
    s[S_DDRA] = e.ddra; s[S_DDRB] = e.ddrb; s[S_SWCHA_OUT] = e.swcha_out; s[S_SWCHB_OUT] = e.swchb_out;
    s[S_CX] = e.cx; s[S_JAM] = e.jam;
    s[S_PADDLE] = v.paddle; s[S_SCORE] = v.score; s[S_TERMINAL] = v.terminal;
    s[S_ALE_LIVES] = v.ale_lives; s[S_STARTED] = v.started; s[S_FRAME_NUMBER] = v.frame_number;
    s[S_LIVES] = v.lives; s[S_WAS_REAL_DONE] = v.was_real_done; s[S_HAS_EPISODE] = v.has_episode;
    s[S_CUR_REWARD] = v.cur_reward; s[S_NUM_STEPS] = v.num_steps; s[S_ELAPSED] = (int)v.elapsed;
    s[S_COMPAT_COUNT] = (int)v.compat_count; s[S_RESET_COUNT] = v.reset_count;
    s[S_OBS_SINGLE] = v.obs_single;
  }
}

DEVI void stage_rom(uint32_t* lds, const uint32_t* romw, int rom_size) {
  for (int i = threadIdx.x; i < rom_size; i += blockDim.x) lds[i] = romw[i];
  __syncthreads();
}

constexpr int kEnvsPerBlock = 4;   // 4 wavefronts share one LDS copy of the cartridge
constexpr int kMaxRomWords = 4096;

enum : int { MODE_STEP = 0, MODE_RESET = 1, MODE_SNAPSHOT = 2 };

// One wavefront per env.  MODE_STEP: VectorEnv.step.  MODE_RESET: VectorEnv.reset.
// MODE_SNAPSHOT: wave k builds reset snapshot k (noops = k+1) for the O(1) real-reset path.
template <int MODE>
__global__ __launch_bounds__(64 * kEnvsPerBlock) void atari_env_kernel(
    uint8_t* __restrict__ states, const uint32_t* __restrict__ romw_g, EnvParams prm,
    const long long* __restrict__ actions, uint8_t* __restrict__ frames,
    float* __restrict__ rewards, uint8_t* __restrict__ dones, uint8_t* __restrict__ obs_flags,
    float* __restrict__ ep_returns, int* __restrict__ ep_lengths,
    uint8_t* __restrict__ snap /* [30][kSnapBytes] or null */, int* __restrict__ jam_out) {
  __shared__ uint32_t rom_lds[kMaxRomWords];
  stage_rom(rom_lds, romw_g, prm.rom_size);
  const int lane = threadIdx.x & 63;
  const int wave = rfl((int)(threadIdx.x >> 6));
  const int e = blockIdx.x * kEnvsPerBlock + wave;
  if (e >= prm.E) return;
  Env v;
  v.emu.romw = rom_lds;
  v.emu.rom_mask = prm.rom_size - 1;
  v.emu.lane = lane;
  v.game = prm.game;
  v.seed = prm.seed;
  v.max_steps = prm.max_episode_steps;
  v.ep_closed = 0; v.ep_return = 0; v.ep_length = 0; v.no_render = 0;
  const bool has_fire = true;  // Pong and Breakout both list FIRE as action 1

  if (MODE == MODE_SNAPSHOT) {
    uint8_t* dst = snap + (size_t)e * kSnapBytes;
    v.buf0 = dst + kStateBytes;
    v.buf1 = v.buf0 + kFrameBytes;
    v.env_id = 0;
    v.emu.system_reset();
    v.emu.romw = rom_lds; v.emu.rom_mask = prm.rom_size - 1; v.emu.lane = lane;
    v.paddle = kPaddleDefault; v.score = v.terminal = v.ale_lives = v.started = v.frame_number = 0;
    v.lives = 0; v.was_real_done = 1; v.has_episode = 0; v.cur_reward = 0; v.num_steps = 0;
    v.elapsed = 0; v.compat_count = 0; v.reset_count = 0; v.obs_single = 0;
    // FireResetEnv.reset with a FIXED noop count e+1 (no RNG draw)
    v.episodic_reset(e + 1, has_fire);
    int r;
    bool d1 = v.episodic_step(ACT_FIRE, r);
    bool d2 = v.episodic_step(v.action_code(2), r);
    // a done inside the canned sequence would need the general path: mark the snapshot unusable
    if (d1 || d2 || v.ep_closed) v.emu.jam |= 0x4000;
    store_env(v, dst, lane);
    return;
  }

  uint8_t* blob = states + (size_t)e * kStateBytes;
  v.buf0 = frames + (size_t)e * 2 * kFrameBytes;
  v.buf1 = v.buf0 + kFrameBytes;
  v.env_id = prm.env_id0 + (unsigned long long)e;

  if (MODE == MODE_RESET) {
    v.emu.system_reset();
    v.emu.romw = rom_lds; v.emu.rom_mask = prm.rom_size - 1; v.emu.lane = lane;
    v.paddle = kPaddleDefault; v.score = v.terminal = v.ale_lives = v.started = v.frame_number = 0;
    v.lives = 0; v.was_real_done = 1; v.has_episode = 0; v.cur_reward = 0; v.num_steps = 0;
    v.elapsed = 0; v.compat_count = 0; v.reset_count = 0; v.obs_single = 0;
    v.fire_reset(has_fire);
    if (lane == 0) obs_flags[e] = (uint8_t)(2 | (v.obs_single ? 1 : 0));
    store_env(v, blob, lane);
    if (lane == 0 && v.emu.jam) atomicOr(jam_out, v.emu.jam);
    return;
  }

  // ---- MODE_STEP ----
  load_env(v, blob, lane);
  v.emu.romw = rom_lds; v.emu.rom_mask = prm.rom_size - 1; v.emu.lane = lane;
  int a = (int)actions[e];
  a = rfl(a);
  if (a < 0 || a >= v.num_actions()) a = 0;
  int total;
  bool done = v.episodic_step(v.action_code(a), total);
  int flags = 0;
  if (done) {
    flags = 2;
    bool fast = false;
    if (snap && v.was_real_done) {
      // O(1) real reset: ALE reset + k noops + the two fire steps are a deterministic function
      // of k, precomputed per k by MODE_SNAPSHOT.  Falls back to the general path when the
      // never-reset CompatWrapper counter could fire inside the sequence.
      uint32_t w[4];
      philox4x32_10(v.seed, (unsigned long long)(unsigned)v.reset_count, v.env_id, w);
      const int k = (int)((uint32_t)rfl((int)w[0]) % 30u);  // noops = k + 1
      const uint8_t* src = snap + (size_t)k * kSnapBytes;
      const int* ss = (const int*)(src + kOffScalars);
      const int delta = rfl(ss[S_COMPAT_COUNT]);
      const int sjam = rfl(ss[S_JAM]);
      if (!(sjam & 0x4000) && v.compat_count + delta < v.max_steps && delta < v.max_steps) {
        fast = true;
        if (v.has_episode) { v.ep_closed++; v.ep_return = v.cur_reward; v.ep_length = v.num_steps; }
        const long long cc = v.compat_count + delta;
        const int rc = v.reset_count + 1;
        const int jam_keep = v.emu.jam;
        load_env(v, src, lane);
        v.emu.romw = rom_lds; v.emu.rom_mask = prm.rom_size - 1; v.emu.lane = lane;
        v.emu.jam |= jam_keep;
        v.compat_count = cc;
        v.reset_count = rc;
        v.has_episode = 1;
        const uint4* fs = (const uint4*)(src + kStateBytes);
        uint4* fd = (uint4*)v.buf0;
        for (int i = lane; i < 2 * kFrameBytes / 16; i += 64) fd[i] = fs[i];
      }
    }
    if (!fast) v.fire_reset(has_fire);
    flags |= v.obs_single ? 1 : 0;
  }
  if (lane == 0) {
    rewards[e] = (float)((total > 0) - (total < 0));  // ClipRewardEnv: np.sign
    dones[e] = done ? 1 : 0;
    obs_flags[e] = (uint8_t)flags;
    ep_returns[e] = (float)v.ep_return;
    ep_lengths[e] = v.ep_closed ? v.ep_length : 0;
    if (v.emu.jam) atomicOr(jam_out, v.emu.jam);
  }
  store_env(v, blob, lane);
}

// ========================================================================================
// frame_post: MaxAndSkipEnv max (atari_wrappers.py:239) + WarpFrame (:263-267) as restated in
// oracle/frame_oracle.c.  One workgroup per env: the two 33,600-byte colour frames are read
// once with 16-byte loads, reduced to max-RGB gray in LDS, then area-resampled from LDS.
// Algorithmic bytes per env-step: 2*33,600 read + dim*dim written.
// ========================================================================================
struct Tap { int si; float alpha; };

DEVI uint32_t gray_of_colors(uint32_t c0, uint32_t c1, const uint32_t* pal) {
  const uint32_t a = pal[c0 >> 1], b = pal[c1 >> 1];
  const uint32_t r0 = (a >> 16) & 255, g0 = (a >> 8) & 255, b0 = a & 255;
  const uint32_t r1 = (b >> 16) & 255, g1 = (b >> 8) & 255, b1 = b & 255;
  const uint32_t r = r0 > r1 ? r0 : r1, g = g0 > g1 ? g0 : g1, bb = b0 > b1 ? b0 : b1;
  return (r * 4899u + g * 9617u + bb * 1868u + 8192u) >> 14;
}

__global__ __launch_bounds__(256) void frame_post_kernel(
    const uint8_t* __restrict__ frames0, const uint8_t* __restrict__ frames1, int64_t in_stride,
    int fmt, const uint8_t* __restrict__ flags, uint8_t* __restrict__ out, int64_t out_stride,
    int dim, const uint8_t* __restrict__ blob) {
  __shared__ uint8_t gray[kFrameBytes];
  __shared__ uint32_t pal[128];
  const int e = blockIdx.x;
  const int* hdr = (const int*)blob;
  const int* xstart = (const int*)(blob + hdr[3]);
  const int* ystart = (const int*)(blob + hdr[4]);
  const Tap* xt = (const Tap*)(blob + hdr[5]);
  const Tap* yt = (const Tap*)(blob + hdr[6]);
  const uint32_t* pal_g = (const uint32_t*)(blob + hdr[7]) - 128;
  if (threadIdx.x < 128) pal[threadIdx.x] = pal_g[threadIdx.x];
  __syncthreads();
  const bool single = (frames1 == nullptr) || (flags && (flags[e] & 1));
  const uint8_t* f0 = frames0 + (size_t)e * in_stride;
  const uint8_t* f1 = single ? f0 : frames1 + (size_t)e * in_stride;
  if (fmt == 1) {
    const uint4* a4 = (const uint4*)f0;
    const uint4* b4 = (const uint4*)f1;
    for (int i = threadIdx.x; i < kFrameBytes / 16; i += blockDim.x) {
      const uint4 a = a4[i], b = b4[i];
      const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
      uint32_t ow[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint32_t o = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          o |= gray_of_colors((aw[q] >> (8 * j)) & 255, (bw[q] >> (8 * j)) & 255, pal) << (8 * j);
        ow[q] = o;
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
  for (int p = threadIdx.x; p < dim * dim; p += blockDim.x) {
    const int dy = p / dim, dx = p - dy * dim;
    const int x0 = xstart[dx], x1 = xstart[dx + 1];
    float sum = 0.f;
    for (int j = ystart[dy]; j < ystart[dy + 1]; ++j) {
      const uint8_t* S = gray + yt[j].si * kW;
      float buf = 0.f;
      for (int k = x0; k < x1; ++k) buf = __fadd_rn(buf, __fmul_rn((float)S[xt[k].si], xt[k].alpha));
      const float tmp = __fmul_rn(yt[j].alpha, buf);
      sum = (j == ystart[dy]) ? tmp : __fadd_rn(sum, tmp);
    }
    // cv::saturate_cast<uchar>(float): cvRound (round half to even) then clamp
    int r = (int)__builtin_rintf(sum);
    r = r < 0 ? 0 : (r > 255 ? 255 : r);
    o[p] = (uint8_t)r;
  }
}

// FrameStack (atari_wrappers.py:270-306) without storing stacks: the rollout ring keeps ONE
// dim*dim frame per (slot, env); a stacked obs is gathered as channel j = ring[slot -
// min(3-j, since)] where `since` = steps since the env's last reset (0 => 4 copies, :290-294).
__global__ __launch_bounds__(256) void stack_gather_kernel(
    const uint8_t* __restrict__ ring, const uint8_t* __restrict__ since, int E, int fsz,
    const int* __restrict__ slots, const int* __restrict__ envs, int64_t n,
    uint8_t* __restrict__ out) {
  // one workgroup per (sample, channel); 16-byte copies
  const int64_t s = blockIdx.x >> 2;
  const int j = blockIdx.x & 3;
  if (s >= n) return;
  const int slot = slots[s];
  const int env = envs[s];
  int back = 3 - j;
  const int sr = since[(size_t)slot * E + env];
  back = back < sr ? back : sr;
  const uint8_t* src = ring + ((size_t)(slot - back) * E + env) * fsz;
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
  return 8 * 4 + 2 * (size_t)(dim + 1) * 4 + (size_t)(nx + ny) * sizeof(Tap) + 128 * 4;
}

PARLHIP_EXPORT int parlhip_frame_post_tables_init(void* host_blob, int dim) {
  if (!host_blob || dim < 1 || dim > 210) return PARLHIP_EINVAL;
  int* hdr = (int*)host_blob;
  int* xstart = hdr + 8;
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
  frame_post_kernel<<<E, 256, 0, (hipStream_t)stream>>>(frames0, frames1, in_stride, fmt, flags, out,
                                                        out_stride, dim, (const uint8_t*)tables_dev);
  return check_launch();
}

PARLHIP_EXPORT size_t parlhip_atari_state_bytes(void) { return kStateBytes; }
PARLHIP_EXPORT size_t parlhip_atari_frame_bytes(void) { return 2 * (size_t)kFrameBytes; }
PARLHIP_EXPORT size_t parlhip_atari_rom_table_bytes(uint32_t rom_size) { return (size_t)rom_size * 4; }
PARLHIP_EXPORT size_t parlhip_atari_reset_cache_bytes(void) { return kNumSnap * kSnapBytes; }

PARLHIP_EXPORT int parlhip_atari_rom_table_build(const uint8_t* rom_host, uint32_t rom_size,
                                                 uint32_t* table_host) {
  if (!rom_host || !table_host) return PARLHIP_EINVAL;
  if (rom_size != 2048 && rom_size != 4096) return PARLHIP_ENOSUP;  // unbanked 2K/4K carts
  build_rom_words(rom_host, rom_size, table_host);
  return PARLHIP_OK;
}

PARLHIP_EXPORT int parlhip_atari_num_actions(int game) {
  return game == GAME_BREAKOUT ? 4 : (game == GAME_PONG ? 6 : -1);
}

static int check_env_args(const void* states, const void* romw, uint32_t rom_size, int game, int E) {
  if (E < 0 || !romw || (E > 0 && !states)) return PARLHIP_EINVAL;
  if (rom_size != 2048 && rom_size != 4096) return PARLHIP_ENOSUP;
  if (game != GAME_PONG && game != GAME_BREAKOUT) return PARLHIP_ENOSUP;
  return PARLHIP_OK;
}

PARLHIP_EXPORT int parlhip_atari_reset_cache_build(const uint32_t* rom_table_dev, uint32_t rom_size,
                                                   int game, int64_t max_episode_steps, void* cache_dev,
                                                   int32_t* jam_flag_dev, parlhip_stream_t stream) {
  int rc = check_env_args((void*)1, rom_table_dev, rom_size, game, 1);
  if (rc) return rc;
  if (!cache_dev || !jam_flag_dev) return PARLHIP_EINVAL;
  EnvParams prm{game, (int)rom_size, kNumSnap, 0ull, 0ull, (long long)max_episode_steps};
  atari_env_kernel<MODE_SNAPSHOT><<<ceil_div(kNumSnap, kEnvsPerBlock), 64 * kEnvsPerBlock, 0,
                                    (hipStream_t)stream>>>(
      nullptr, rom_table_dev, prm, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
      (uint8_t*)cache_dev, jam_flag_dev);
  return check_launch();
}

PARLHIP_EXPORT int parlhip_atari_vec_reset(void* states, const uint32_t* rom_table_dev, uint32_t rom_size,
                                           int game, uint8_t* frames, uint8_t* obs_flags, int E,
                                           uint64_t seed, uint64_t env_id0, int64_t max_episode_steps,
                                           int32_t* jam_flag_dev, parlhip_stream_t stream) {
  int rc = check_env_args(states, rom_table_dev, rom_size, game, E);
  if (rc) return rc;
  if (E == 0) return PARLHIP_OK;
  if (!frames || !obs_flags || !jam_flag_dev) return PARLHIP_EINVAL;
  EnvParams prm{game, (int)rom_size, E, seed, env_id0, (long long)max_episode_steps};
  atari_env_kernel<MODE_RESET><<<ceil_div(E, kEnvsPerBlock), 64 * kEnvsPerBlock, 0, (hipStream_t)stream>>>(
      (uint8_t*)states, rom_table_dev, prm, nullptr, frames, nullptr, nullptr, obs_flags, nullptr, nullptr,
      nullptr, jam_flag_dev);
  return check_launch();
}

PARLHIP_EXPORT int parlhip_atari_vec_step(void* states, const uint32_t* rom_table_dev, uint32_t rom_size,
                                          int game, const int64_t* actions, uint8_t* frames, float* rewards,
                                          uint8_t* dones, uint8_t* obs_flags, float* ep_returns,
                                          int32_t* ep_lengths, int E, uint64_t seed, uint64_t env_id0,
                                          int64_t max_episode_steps, const void* reset_cache_dev,
                                          int32_t* jam_flag_dev, parlhip_stream_t stream) {
  int rc = check_env_args(states, rom_table_dev, rom_size, game, E);
  if (rc) return rc;
  if (E == 0) return PARLHIP_OK;
  if (!actions || !frames || !rewards || !dones || !obs_flags || !ep_returns || !ep_lengths || !jam_flag_dev)
    return PARLHIP_EINVAL;
  EnvParams prm{game, (int)rom_size, E, seed, env_id0, (long long)max_episode_steps};
  atari_env_kernel<MODE_STEP><<<ceil_div(E, kEnvsPerBlock), 64 * kEnvsPerBlock, 0, (hipStream_t)stream>>>(
      (uint8_t*)states, rom_table_dev, prm, (const long long*)actions, frames, rewards, dones, obs_flags,
      ep_returns, ep_lengths, (uint8_t*)reset_cache_dev, jam_flag_dev);
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
                                                                          envs, n, out);
  return check_launch();
}
