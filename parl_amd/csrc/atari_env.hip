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
//
// The wrappers are recursive Python; here they are flattened into a small per-wave state
// machine whose ONLY action is "emulate one more frame with this input, into that buffer" so
// that the 6507/TIA code exists once in the kernel (instruction-cache footprint).
#include "common.hpp"
#ifdef PARLHIP_ENV_ENTRYHIST  // diagnostic build only (tools/env_entry_hist.py): clocks / 6507 instructions / entries of the translated code by entry PC
namespace parlhip { namespace atari { __device__ unsigned long long g_env_entryhist[8192][3]; } }
#define PARLHIP_ENTRY_HIST(pc, clocks, instrs)                                                  \
  do {                                                                                          \
    if (lane == 0) {                                                                            \
      const int k_ = (pc) & 8191;                                                               \
      atomicAdd(&g_env_entryhist[k_][0], (unsigned long long)(clocks));                         \
      atomicAdd(&g_env_entryhist[k_][1], (unsigned long long)(instrs));                         \
      atomicAdd(&g_env_entryhist[k_][2], 1ull);                                                 \
    }                                                                                           \
  } while (0)
#endif
#include "atari_core.hpp"
#include "frame_tail.hpp"
#include "philox.hpp"
#include "policy_head.hpp"

namespace parlhip {
namespace atari {

constexpr int kNumSnap = 30;  // NoopResetEnv noop_max
constexpr size_t kSnapBytes = kStateBytes + 2 * (size_t)kFrameBytes;  // 67,712 (16-B multiple)
static_assert(kSnapBytes % 16 == 0, "snapshot stride must keep 16-byte alignment");

struct EnvParams {
  int game, rom_size, E, mode;
  unsigned long long seed, env_id0;
  long long max_episode_steps;
  int budget;  // elastic stepping: frames one launch may emulate per env (0: every step runs to its end)
};

// What a launch does besides VectorEnv.step itself (parlhip_atari_vec_step_obs).  It is the FIRST kernel argument
// and the kernel body never names it: the two ends of the launch read it through the kernarg segment pointer where
// they need it (fuse_args()).  A named argument is loaded in the entry block and stays live across wave A's frame
// loop — whose register allocation does not forgive that (six pointers as plain arguments once took the kernel from
// 420 to 24,881 SGPR spills, see elastic stepping below); the kernarg pointer is one SGPR pair.
struct StepFuse {
  // tail: the observation (frame_tail.hpp).  obs_out == nullptr: the colour frames stay raw and a
  // parlhip_frame_post_*_u8 launch follows
  uint8_t* obs_out;           // [E, dim * dim]: the ring slot of this step
  const uint8_t* tables;      // parlhip_frame_post_tables_init blob
  const uint8_t* since_prev;  // [E] FrameStack counters of the previous slot, or null (= 0)
  uint8_t* since_next;        // [E]
  double* ep_acc;             // [3] MonitorEnv sums (parlhip_episode_stats_accum_f64), or null
  int dim, A;
  // head: the actors' policy head + draw (policy_head.hpp).  hidden == nullptr: the actions are read from `actions`
  const float* hidden;        // [E, 256] trunk output of the actors' model for the observation the envs hold
  const float* w_pi;          // [A, 256]
  const float* b_pi;          // [A]
  float* logits_out;          // [E, A]: behaviour logits row of this step's slab
  long long* actions_out;     // [E]
  const unsigned long long* offset_base;  // or null: added to `offset` on the device (hipGraph replays)
  unsigned long long sample_seed, offset, row0;   // Philox key / counter of the draw: (offset, row0 + e)
};
DEVI const StepFuse* fuse_args() { return (const StepFuse*)__builtin_amdgcn_kernarg_segment_ptr(); }

// The actors' policy head + draw at the HEAD of an env's CPU wave (parlhip_atari_vec_step_policy_obs): what
// policy_head_sample_kernel does in a launch of its own between the actors' last GEMM and the emulator — 8-13 us
// and a launch gap per env step for 1,800 multiply-adds per env — with the same instructions (policy_head.hpp).
// Every lane ends with the action; lane 0 stores the logits row and the action into the rollout slabs.
__device__ __attribute__((noinline)) int policy_head_main(const StepFuse* fz_v, int e_v) {
  const unsigned long long x = (unsigned long long)(uintptr_t)fz_v;
  const StepFuse* fz = (const StepFuse*)(uintptr_t)(((unsigned long long)(uint32_t)rfl((int)(x >> 32)) << 32) |
                                                     (uint32_t)rfl((int)(uint32_t)x));
  const int e = rfl(e_v), lane = (int)(threadIdx.x & 63);
  const int A = fz->A;
  unsigned long long offset = fz->offset;
  const unsigned long long* ob = fz->offset_base;
  if (ob) offset += *ob;
  float row[6];
  policy_head_row<6>(fz->hidden + (size_t)e * 256, fz->w_pi, fz->b_pi, A, lane, row);
  const double u = philox_uniform53(fz->sample_seed, offset, fz->row0 + (unsigned long long)e);
  const long long a = policy_draw<6>(row, A, u);
  if (lane == 0) {
    float* lo = fz->logits_out + (size_t)e * A;
#pragma unroll
    for (int k = 0; k < 6; ++k) if (k < A) lo[k] = row[k];
    fz->actions_out[e] = a;
  }
  return (int)a;
}

// Elastic stepping (parlhip_atari_vec_step_elastic): a launch emulates at most `budget` frames per
// env.  An env whose step needs more (the 12 frames of a life-loss reset: EpisodicLifeEnv's NOOP step +
// FireResetEnv's two steps; the 64+ frames of a real reset when the snapshot cache cannot be used)
// parks its wrapper state machine in the state blob (S_SUSP*) and goes on in the next launches,
// taking no action and delivering no observation until its step is complete; the others keep
// stepping.  Each env still sees exactly the sequence of frames and inputs the synchronous path gives
// it.  The per-batch row accounting lives in two tiny kernels around the emulator (elastic_pre /
// elastic_post): anything more that is live across the emulator's frame loop costs it dearly — with
// the six row pointers as arguments of the env kernel itself its SGPR spills went from 420 to 24,881.
enum : int { CTL_STEP = 0, CTL_CONTINUE = 1, CTL_IDLE = 2 };

enum : int { MODE_STEP = 0, MODE_RESET = 1, MODE_SNAPSHOT = 2 };

// state-machine phases: what the frame being emulated belongs to
enum : int {
  PH_SKIP = 0,  // a MaxAndSkipEnv.step frame (ctx says which caller)
  PH_ALE,       // ALE reset_game(): 60 NOOP + 4 RESET-switch frames, no wrapper accounting
  PH_NOOP,      // NoopResetEnv.reset noop frame
  PH_END
};
enum : int { CTX_MAIN = 0, CTX_FIRE1, CTX_FIRE2, CTX_LIFE };  // who called MaxAndSkipEnv.step
enum : int { TO_B = 0, TO_C, TO_END };                        // continuation after episodic_reset

constexpr int kEnvsPerBlock = 4;   // 4 envs share one LDS copy of the cartridge: 8 wavefronts, a CPU wave and a picture wave per env
constexpr int kMaxRomWords = 4096;

DEVI int action_code(int idx) {  // ALE minimal action sets (Pong 6, Breakout the first 4)
  return idx == 0 ? ACT_NOOP : idx == 1 ? ACT_FIRE : idx == 2 ? ACT_RIGHT : idx == 3 ? ACT_LEFT
         : idx == 4 ? ACT_RIGHTFIRE : ACT_LEFTFIRE;
}

// ---- state blob <-> registers ----
struct Wrap {  // ALE + wrapper state (wave-uniform)
  int paddle, score, terminal, ale_lives, started, frame_number;
  int lives, was_real_done, has_episode, cur_reward, num_steps, reset_count, obs_single;
  long long elapsed, compat_count;
};

DEVI void load_env(Emu& e, Wrap& v, const uint8_t* blob, int lane) {
  const int* s = (const int*)(blob + kOffScalars);
  e.ram_lo = blob[kOffRam + lane];
  e.ram_hi = blob[kOffRam + 64 + lane];
  e.tiac = blob[kOffTia + lane];   // the CPU-side register file; the render-side one is wave B's
  auto L = [&](int i) { return rfl(s[i]); };
  e.A = L(S_A); e.X = L(S_X); e.Y = L(S_Y); e.S = L(S_S); e.pset(L(S_P)); e.PC = L(S_PC);
  e.cyc = L(S_CYC); e.cyc0 = L(S_CYC0);
  e.vsync_finish = L(S_VSYNC_FINISH); e.dump_dis_cyc = L(S_DUMP_DIS_CYC); e.dump_en = L(S_DUMP_EN);
  e.timer = L(S_TIMER); e.timer_shift = L(S_TIMER_SHIFT); e.timer_set_cyc = L(S_TIMER_SET_CYC);
  e.ddra = L(S_DDRA); e.ddrb = L(S_DDRB); e.swcha_out = L(S_SWCHA_OUT); e.swchb_out = L(S_SWCHB_OUT);
  e.cx = 0; e.jam = L(S_JAM); e.stop = 0;   // (collision latches: wave B's; read through a SYNC record)
  e.pneed0 = e.pneed1 = Emu::paddle_needed(kPaddleDefault); e.fire0 = e.fire1 = e.sw_reset = 0;
  e.fb = nullptr;
  v.paddle = L(S_PADDLE); v.score = L(S_SCORE); v.terminal = L(S_TERMINAL);
  v.ale_lives = L(S_ALE_LIVES); v.started = L(S_STARTED); v.frame_number = L(S_FRAME_NUMBER);
  v.lives = L(S_LIVES); v.was_real_done = L(S_WAS_REAL_DONE); v.has_episode = L(S_HAS_EPISODE);
  v.cur_reward = L(S_CUR_REWARD); v.num_steps = L(S_NUM_STEPS); v.elapsed = L(S_ELAPSED);
  v.compat_count = L(S_COMPAT_COUNT); v.reset_count = L(S_RESET_COUNT);
  v.obs_single = L(S_OBS_SINGLE);
}

DEVI void store_env(const Emu& e, const Wrap& v, uint8_t* blob, int lane) {
  int* s = (int*)(blob + kOffScalars);
  blob[kOffRam + lane] = (uint8_t)e.ram_lo;
  blob[kOffRam + 64 + lane] = (uint8_t)e.ram_hi;
  // (TIA register bytes, collision latches, picture position: wave B stores them, Emu::render_main)
  if (lane == 0) {
    s[S_A] = e.A; s[S_X] = e.X; s[S_Y] = e.Y; s[S_S] = e.S; s[S_P] = e.pfull(); s[S_PC] = e.PC;
    s[S_BUS] = 0;
    s[S_CYC] = e.cyc; s[S_CYC0] = e.cyc0;
    s[S_VSYNC_FINISH] = e.vsync_finish; s[S_DUMP_DIS_CYC] = e.dump_dis_cyc; s[S_DUMP_EN] = e.dump_en;
    s[S_TIMER] = e.timer; s[S_TIMER_SHIFT] = e.timer_shift; s[S_TIMER_SET_CYC] = e.timer_set_cyc;
    s[S_DDRA] = e.ddra; s[S_DDRB] = e.ddrb; s[S_SWCHA_OUT] = e.swcha_out; s[S_SWCHB_OUT] = e.swchb_out;
    s[S_JAM] = e.jam;
    s[S_PADDLE] = v.paddle; s[S_SCORE] = v.score; s[S_TERMINAL] = v.terminal;
    s[S_ALE_LIVES] = v.ale_lives; s[S_STARTED] = v.started; s[S_FRAME_NUMBER] = v.frame_number;
    s[S_LIVES] = v.lives; s[S_WAS_REAL_DONE] = v.was_real_done; s[S_HAS_EPISODE] = v.has_episode;
    s[S_CUR_REWARD] = v.cur_reward; s[S_NUM_STEPS] = v.num_steps; s[S_ELAPSED] = (int)v.elapsed;
    s[S_COMPAT_COUNT] = (int)v.compat_count; s[S_RESET_COUNT] = v.reset_count;
    s[S_OBS_SINGLE] = v.obs_single;
  }
}

#ifdef PARLHIP_ENV_TIMING  // diagnostic build only (tools/env_wave_times.py): per-wave start / end clocks
__device__ unsigned long long g_env_t0[8192], g_env_t1[8192];
#endif
#ifdef PARLHIP_ENV_REGIONS  // diagnostic build only (tools/env_regions.py): where a wave's launch goes
__device__ unsigned long long g_env_regions[8192][24];
#endif
#ifdef PARLHIP_ENV_TRACEITER
__device__ unsigned long long g_env_traceiter[8192][16];
#endif

// The env's picture: render-side TIA registers, collision latches, frame buffers (Emu::render_main).  NOT inlined
// into the env kernel: a function of its own is register-allocated on its own — inside the kernel it shared one
// allocation with the translated cartridges (the biggest function of the library, 500-1500 SGPR spills), and the
// sixteen scalar registers of its register file went to spill lanes.  One call per launch; the arguments arrive in
// vector registers (the calling convention) and are made wave-uniform again.
__device__ __attribute__((noinline)) int picture_wave_main(uint8_t* blob_v, const uint8_t* snap_v, RenderQueue* rq_v, int e_v,
                                                           const uint8_t* frames_v, uint32_t* lds_hi_v, int wave_v,
                                                           const StepFuse* fz_v) {
  auto uni = [](const void* p) -> unsigned long long {
    const unsigned long long x = (unsigned long long)(uintptr_t)p;
    return ((unsigned long long)(uint32_t)rfl((int)(x >> 32)) << 32) | (uint32_t)rfl((int)(uint32_t)x);
  };
  uint8_t* blob = (uint8_t*)(uintptr_t)uni(blob_v);
  const uint8_t* snap = (const uint8_t*)(uintptr_t)uni(snap_v);
  Emu r;
  r.lane = (int)(threadIdx.x & 63);
  r.rq = (LdsRenderQueue*)(RenderQueue*)(uintptr_t)uni(rq_v);   // generic -> LDS address space
#ifdef PARLHIP_ENV_REGIONS
  for (int i = 0; i < 5; ++i) { r.rt[i] = 0; r.rn[i] = 0; }
#endif
  // bands [b0, b1) of the observation of the frame being drawn (Emu::fb_flags): only launches that deliver observations
  // ever flag a frame
  auto convert = [&](int b0, int b1) {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // this wave's own frame stores, before it reads them back
    const StepFuse* fz = (const StepFuse*)(uintptr_t)uni(fz_v);   // (the kernarg segment pointer is the KERNEL's: an argument here)
    const int dim = fz->dim, env = rfl(e_v);
    obs_tail_dispatch((const uint8_t*)(uintptr_t)uni(frames_v), fz->obs_out + (size_t)env * dim * dim, fz->tables,
                      (uint32_t*)(uintptr_t)uni(lds_hi_v), dim, b0, b1, 0, rfl(wave_v));
  };
  const uint32_t exit_w0 = r.render_main(blob, snap, kSnapBytes, convert);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if ((exit_w0 & 0x300u) == 0x300u) {
    // wave A's last frame was the flagged one and it is drawn as far as it will ever be: every band is final; both
    // waves claim chunks until none is left
    __atomic_store_n(&r.rq->obs_ready, (uint32_t)kObsBands, __ATOMIC_RELAXED);
    for (;;) {
      if (r.obs_cur >= r.obs_end) r.obs_take();
      if (r.obs_cur >= kObsBands) break;
      convert(r.obs_cur, r.obs_end);
      r.obs_cur = r.obs_end;
    }
  }
  // the launch's last frame is drawn and stored: the env's CPU wave may read the frame pair (the observation's two chunks
  // when the flagged frame was not the last one)
  __atomic_store_n(&r.rq->fin, 1u, __ATOMIC_RELAXED);
#ifdef PARLHIP_ENV_REGIONS
  const int e = rfl(e_v);
  if (r.lane == 0 && e < 8192) { g_env_regions[e][9] = r.rt[3]; g_env_regions[e][10] = r.rt[4]; g_env_regions[e][11] = r.rt[0];
    g_env_regions[e][12] = r.rt[1]; g_env_regions[e][13] = (unsigned long long)r.rn[0]; g_env_regions[e][14] = (unsigned long long)r.rn[1];
    g_env_regions[e][20] = (unsigned long long)r.rn[4]; g_env_regions[e][17] = r.rt[2]; g_env_regions[e][18] = (unsigned long long)r.rn[2]; g_env_regions[e][19] = __builtin_readcyclecounter(); }
#else
  (void)e_v;
#endif
  return (int)exit_w0;
}

// Two wavefronts per env (atari_core.hpp): waves 0 .. 3 of a workgroup run the 6507 / RIOT / wrapper chain of its
// four envs, waves 4 .. 7 their pictures.  mode STEP: VectorEnv.step; RESET: VectorEnv.reset; SNAPSHOT: env k
// builds reset snapshot k (noops = k+1) for the O(1) real-reset path.
// __launch_bounds__(512, 4 waves per SIMD) = at most 128 VGPRs (the kernel needed 160 / 139 unconstrained; no scratch
// at 128): an env workgroup leaves half of every SIMD's register file to the learner's MFMA kernels (~200 VGPRs per
// wave), which otherwise cannot be resident beside it at all — and since the emulator is on every CU for 80 % of an
// env step, the learner then only ran in the gaps and the pipeline became learner-bound (measured: the actors idle
// 9 ms of every 50 ms rollout waiting for the learner's pass; tools/actor_gaps.sh).
template <int GAME>
__global__ __launch_bounds__(128 * kEnvsPerBlock, 4) void atari_env_kernel(
    StepFuse /* read through fuse_args(), see StepFuse */, uint8_t* __restrict__ states, const uint32_t* __restrict__ romw_g, EnvParams prm,
    const long long* __restrict__ actions, uint8_t* __restrict__ frames,
    float* __restrict__ rewards, uint8_t* __restrict__ dones, uint8_t* __restrict__ obs_flags,
    float* __restrict__ ep_returns, int* __restrict__ ep_lengths,
    uint8_t* __restrict__ snap /* [30][kSnapBytes] or null */, int* __restrict__ jam_out,
    const uint8_t* __restrict__ ctl /* [E] CTL_* per env, or null */) {
  __shared__ uint32_t rom_lds[kMaxRomWords];
  __shared__ RenderQueue rqs[kEnvsPerBlock];
  for (int i = threadIdx.x; i < prm.rom_size; i += blockDim.x) rom_lds[i] = romw_g[i];
  if (threadIdx.x < kEnvsPerBlock) { rqs[threadIdx.x].wr = 0; rqs[threadIdx.x].rd = 0; rqs[threadIdx.x].cx = 0; rqs[threadIdx.x].fin = 0;
                                    rqs[threadIdx.x].obs_next = 0; rqs[threadIdx.x].obs_ready = 0; }
  // the observation tail's two colour tables, in the half of rom_lds a 2K cartridge leaves free
  if (fuse_args()->obs_out) obs_tail_stage_tables(rom_lds + kMaxRomWords / 2, fuse_args()->tables, threadIdx.x);
  __syncthreads();
  // One wavefront per env is a long serial dependency chain: when other kernels (the learner's
  // GEMMs on another stream) share the SIMD, this wave should win every issue arbitration.
#ifndef PARLHIP_ENV_PRIO
#define PARLHIP_ENV_PRIO 3
#endif
  __builtin_amdgcn_s_setprio(PARLHIP_ENV_PRIO);
  const int lane = threadIdx.x & 63;
  const int wave = rfl((int)(threadIdx.x >> 6));
  const int slot = wave & (kEnvsPerBlock - 1);
  const bool picture_wave = wave >= kEnvsPerBlock;
  const int e = blockIdx.x * kEnvsPerBlock + slot;
  if (e >= prm.E) return;
  // elastic stepping: this env's rows of the batch are complete, it waits for the others (its obs_flags /
  // ep_lengths were written by elastic_pre_kernel).  The exit must be HERE: the same `return` placed
  // after the Emu state exists (inside the MODE_STEP arm) took the kernel from 460 to 24,053 SGPR
  // spills and its compile from 15 s to 190 s.
  if (ctl && rfl((int)ctl[e]) == CTL_IDLE) return;
#ifdef PARLHIP_ENV_TIMING
  if (lane == 0 && e < 8192 && !picture_wave) g_env_t0[e] = wall_clock64();
#endif
  const int mode = prm.mode, game = prm.game;
  if (picture_wave) {
    uint8_t* rblob = mode == MODE_SNAPSHOT ? snap + (size_t)e * kSnapBytes : states + (size_t)e * kStateBytes;
    const int exit_w0 = rfl(picture_wave_main(rblob, snap, &rqs[slot], e, frames + (size_t)e * 2 * kFrameBytes,
                                              rom_lds + kMaxRomWords / 2, wave, fuse_args()));
    if ((exit_w0 & 0x300) == 0x100) {   // wave A asks for the observation of another frame pair than the flagged one: this wave takes the second chunk of bands
      const StepFuse* fz = fuse_args();
      const int dim = fz->dim;
      obs_tail_dispatch(frames + (size_t)e * 2 * kFrameBytes, fz->obs_out + (size_t)e * dim * dim, fz->tables,
                        rom_lds + kMaxRomWords / 2, dim, kObsFirst, kObsBands, exit_w0 & 1, wave);
    }
    return;
  }
  const bool native_ok = NativeCart<GAME>::present && rfl((int)(rom_lds[0] >> 28)) == GAME;
  const long long max_steps = prm.max_episode_steps;

  Emu emu;
  Wrap v;
#ifdef PARLHIP_ENV_REGIONS
  const unsigned long long k0 = __builtin_readcyclecounter();
  for (int i = 0; i < 5; ++i) { emu.rt[i] = 0; emu.rn[i] = 0; }
#endif
#ifdef PARLHIP_ENV_TRACEITER
  emu.ti_sum = emu.ti_cnt = emu.ti_other = 0;
  emu.ti_last = 0;
#endif
#ifdef PARLHIP_ROM_SCALAR
  emu.romw = (Emu::RomWords)romw_g;
#else
  emu.romw = rom_lds;
#endif
  emu.rom_mask = prm.rom_size - 1;
  emu.lane = lane;
  emu.rq = (LdsRenderQueue*)&rqs[slot];
  emu.rq_wr = 0;
  emu.cx_spec = 0;
  emu.cx_spec_seq = 0;
  emu.fb_flags = 0;
  emu.wqn = 0;
  emu.wq = emu.wq2 = 0;
  uint8_t* blob;
  uint8_t *buf0, *buf1;
  unsigned long long env_id;
  if (mode == MODE_SNAPSHOT) {
    blob = snap + (size_t)e * kSnapBytes;
    buf0 = blob + kStateBytes;
    env_id = 0;
  } else {
    blob = states + (size_t)e * kStateBytes;
    buf0 = frames + (size_t)e * 2 * kFrameBytes;
    env_id = prm.env_id0 + (unsigned long long)e;
  }
  buf1 = buf0 + kFrameBytes;

  // ---- machine registers ----
  int phase, ctx = CTX_MAIN, cont = TO_B, skip_i = 0, skip_total = 0, skip_act = ACT_NOOP;
  int ale_j = 0, noops_left = 0, no_render = 0;
  int ep_closed = 0, ep_return = 0, ep_length = 0;
  int out_total = 0, out_done = 0, did_reset = 0;
  const int fixed_noops = mode == MODE_SNAPSHOT ? e + 1 : 0;

  // frames this launch may still emulate (a counter, not a loop-invariant `budget &&` test: LLVM
  // would unswitch the frame loop on it, i.e. duplicate the emulator)
  int frames_left = (mode == MODE_STEP && prm.budget) ? prm.budget : 0x7fffffff;
  if (mode == MODE_STEP) {
    const int c = ctl ? rfl((int)ctl[e]) : CTL_STEP;
    load_env(emu, v, blob, lane);
    if (c == CTL_CONTINUE) {  // go on where the previous launch stopped; no action is consumed
      const int* sc = (const int*)(blob + kOffScalars);
      const int susp = rfl(sc[S_SUSP]);
      phase = susp & 3; ctx = (susp >> 2) & 3; cont = (susp >> 4) & 3; skip_i = (susp >> 6) & 7;
      no_render = (susp >> 9) & 1;
      skip_total = rfl(sc[S_SUSP_TOTAL]); skip_act = rfl(sc[S_SUSP_ACT]);
      ale_j = rfl(sc[S_SUSP_ALE_J]); noops_left = rfl(sc[S_SUSP_NOOPS]);
      did_reset = 1;  // only reset sequences are ever cut
    } else {
      const StepFuse* fzh = fuse_args();
      int a = fzh->hidden ? rfl(policy_head_main(fzh, e)) : rfl((int)actions[e]);
      const int na = game == GAME_BREAKOUT ? 4 : 6;
      if (a < 0 || a >= na) a = 0;
      phase = PH_SKIP; ctx = CTX_MAIN; skip_act = action_code(a);
    }
  } else {
    emu.system_reset();
    v.paddle = kPaddleDefault; v.score = v.terminal = v.ale_lives = v.started = v.frame_number = 0;
    v.lives = 0; v.was_real_done = 1; v.has_episode = 0; v.cur_reward = 0; v.num_steps = 0;
    v.elapsed = 0; v.compat_count = 0; v.reset_count = 0; v.obs_single = 0;
    // FrameStack.reset -> FireResetEnv.reset -> EpisodicLifeEnv.reset (real) -> NoopResetEnv.reset
    did_reset = 1;
    phase = PH_ALE; ale_j = 0; cont = TO_B;
  }

  // Transitions: none of them emulates a frame — they only choose the next phase.  Macros, not
  // lambdas: a by-reference closure that is not inlined forces the machine registers into
  // scratch memory, and private-memory loads are "divergent" to the compiler — the whole 6507
  // then gets compiled to VALU code.
#define BEGIN_MONITOR_RESET() do { phase = PH_ALE; ale_j = 0; } while (0)
#define BEGIN_SKIP(c, a) do { phase = PH_SKIP; ctx = (c); skip_act = (a); skip_i = 0; skip_total = 0; } while (0)
  // what follows EpisodicLifeEnv.reset inside FireResetEnv.reset (lives: atari_wrappers.py:210)
#define CONTINUE_AFTER() do {                                              \
    v.lives = v.ale_lives;                                                 \
    if (cont == TO_B) BEGIN_SKIP(CTX_FIRE1, ACT_FIRE);                     \
    else if (cont == TO_C) BEGIN_SKIP(CTX_FIRE2, action_code(2));          \
    else { no_render = 0; v.obs_single = 0; phase = PH_END; }              \
  } while (0)
  // EpisodicLifeEnv.reset :200-211
#define BEGIN_EPISODIC_RESET(c) do {                                       \
    cont = (c);                                                            \
    if (v.was_real_done) BEGIN_MONITOR_RESET();                            \
    else BEGIN_SKIP(CTX_LIFE, ACT_NOOP);                                   \
  } while (0)

  while (phase != PH_END) {
    if (frames_left == 0) break;  // suspended: phase != PH_END at the exit
    frames_left--;
    // ------------------------------------------------------------------ choose input + target
    int act;
    uint8_t* fbp = nullptr;
    if (phase == PH_SKIP) {
      act = skip_act;
      if (!no_render) fbp = skip_i == 2 ? buf0 : (skip_i == 3 ? buf1 : nullptr);
    } else if (phase == PH_ALE) {
      if (ale_j == 0) {  // ALE reset_game(): system reset first
        v.paddle = kPaddleDefault;
        const int keep = emu.jam;
        emu.system_reset();
        emu.jam = keep;
      }
      act = ale_j < 60 ? ACT_NOOP : ACT_RESET;
    } else {  // PH_NOOP
      act = ACT_NOOP;
    }
    {  // ALEState::applyActionPaddles
      int delta = 0, fire = 0;
      emu.sw_reset = act == ACT_RESET;
      if (act == ACT_RIGHT || act == ACT_RIGHTFIRE) delta = -kPaddleDelta;
      if (act == ACT_LEFT || act == ACT_LEFTFIRE) delta = kPaddleDelta;
      if (act == ACT_FIRE || act == ACT_RIGHTFIRE || act == ACT_LEFTFIRE) fire = 1;
      v.paddle += delta;
      v.paddle = v.paddle < kPaddleMin ? kPaddleMin : (v.paddle > kPaddleMax ? kPaddleMax : v.paddle);
      const bool swap = game == GAME_PONG;  // Stella props: Video Olympics SwapPaddles=YES
      emu.pneed0 = Emu::paddle_needed(swap ? kPaddleDefault : v.paddle);
      emu.pneed1 = Emu::paddle_needed(swap ? v.paddle : kPaddleDefault);
      emu.fire0 = swap ? 0 : fire;
      emu.fire1 = swap ? fire : 0;
    }
    // ------------------------------------------------------------------ THE frame
#ifdef PARLHIP_ENV_REGIONS
    if (emu.rt[3] == 0) emu.rt[3] = __builtin_readcyclecounter() - k0;   // launch start -> first frame (staging, load, policy head)
#endif
    // the 4th frame of the agent step whose observation this launch delivers: unless the episode ends in it, it is the
    // launch's last (obs_single is 0 after a completed step)
    emu.fb_flags = (mode == MODE_STEP && phase == PH_SKIP && ctx == CTX_MAIN && skip_i == 3 && fuse_args()->obs_out != nullptr) ? 1 : 0;
    emu.frame<GAME>(fbp, native_ok);
    // ------------------------------------------------------------------ after the frame
    if (phase == PH_ALE) {
      ale_j++;
      if (ale_j == 64) {
        // RomSettings::reset + the rest of MonitorEnv.reset / TimeLimit.reset
        v.score = 0; v.terminal = 0; v.started = 0;
        v.ale_lives = game == GAME_BREAKOUT ? 5 : 0;
        v.frame_number = 0;
        v.elapsed = 0;
        if (v.has_episode) { ep_closed++; ep_return = v.cur_reward; ep_length = v.num_steps; }
        v.has_episode = 1;
        v.cur_reward = 0;
        v.num_steps = 0;
        if (noops_left > 0) {
          // monitor_reset fired from inside the noop loop (:127-128): the loop just continues
          noops_left--;
          if (noops_left == 0) { v.obs_single = 1; CONTINUE_AFTER(); }
          else phase = PH_NOOP;
        } else {
          int n;
          if (fixed_noops) {
            n = fixed_noops;
          } else {  // np_random.randint(1, 31) restated: philox(seed; reset_count, env_id)
            uint32_t w[4];
            philox4x32_10(prm.seed, (unsigned long long)(unsigned)v.reset_count, env_id, w);
            v.reset_count++;
            n = 1 + (int)((uint32_t)rfl((int)w[0]) % 30u);
          }
          noops_left = n;
          phase = PH_NOOP;
        }
      }
      continue;
    }
    // raw_step accounting: RomSettings::step, TimeLimit, CompatWrapper, MonitorEnv
    int reward = 0;
    if (game == GAME_PONG) {
      const int x = emu.ram_rd(13), y = emu.ram_rd(14);
      const int sc = y - x;
      reward = sc - v.score;
      v.score = sc;
      v.terminal = (x == 21 || y == 21);
      v.ale_lives = 0;
    } else {
      const int x = emu.ram_rd(77), y = emu.ram_rd(76);
      const int sc = (x & 0x0f) + 10 * ((x & 0xf0) >> 4) + 100 * (y & 0x0f);
      reward = sc - v.score;
      v.score = sc;
      const int lv = emu.ram_rd(57);
      if (!v.started && lv == 5) v.started = 1;
      v.terminal = v.started && lv == 0;
      v.ale_lives = lv;
    }
    v.frame_number++;
    bool done = v.terminal != 0;
    v.elapsed++;
    if (v.elapsed >= max_steps) done = true;
    v.compat_count++;
    if (v.compat_count >= max_steps) { done = true; v.compat_count = 0; }
    v.cur_reward += reward;
    v.num_steps++;

    if (phase == PH_NOOP) {
      if (done) {
        BEGIN_MONITOR_RESET();  // noops_left stays > 0: resume the loop after the ALE reset
      } else {
        noops_left--;
        if (noops_left == 0) { v.obs_single = 1; CONTINUE_AFTER(); }
      }
      continue;
    }
    // PH_SKIP
    skip_total += reward;
    skip_i++;
    if (!done && skip_i < 4) continue;
    v.obs_single = 0;
    if (ctx == CTX_LIFE) {  // the NOOP step of a non-real EpisodicLifeEnv.reset: result ignored
      CONTINUE_AFTER();
      continue;
    }
    // EpisodicLifeEnv.step :186-198
    v.was_real_done = done;
    bool d = done;
    if (v.ale_lives < v.lives && v.ale_lives > 0) d = true;
    v.lives = v.ale_lives;
    if (ctx == CTX_MAIN) {
      out_total = skip_total;
      out_done = d;
      if (!d) { phase = PH_END; continue; }
      did_reset = 1;
      if (snap && mode == MODE_STEP && v.was_real_done) {
        // O(1) real reset: ALE reset + k noops + the two fire steps are a deterministic function
        // of k, precomputed per k by MODE_SNAPSHOT.  Falls back to the general path when the
        // never-reset CompatWrapper counter could fire inside the sequence.
        uint32_t w[4];
        philox4x32_10(prm.seed, (unsigned long long)(unsigned)v.reset_count, env_id, w);
        const int k = (int)((uint32_t)rfl((int)w[0]) % 30u);  // noops = k + 1
        const uint8_t* src = snap + (size_t)k * kSnapBytes;
        const int* ss = (const int*)(src + kOffScalars);
        const int delta = rfl(ss[S_COMPAT_COUNT]);
        const int sjam = rfl(ss[S_JAM]);
        if (!(sjam & 0x4000) && v.compat_count + delta < max_steps && delta < max_steps) {
          if (v.has_episode) { ep_closed++; ep_return = v.cur_reward; ep_length = v.num_steps; }
          const long long cc = v.compat_count + delta;
          const int rc = v.reset_count + 1;
          const int jam_keep = emu.jam;
          emu.rq_wait_idle();  // the picture wave may still be drawing into the frame pair this restore overwrites
          load_env(emu, v, src, lane);
          emu.rq_ctl(Emu::LA_RELOAD, (uint32_t)k, 0);
          emu.cx_spec = 0;
          emu.jam |= jam_keep;
          v.compat_count = cc;
          v.reset_count = rc;
          v.has_episode = 1;
          const uint4* fs = (const uint4*)(src + kStateBytes);
          uint4* fd = (uint4*)buf0;
          for (int i = lane; i < 2 * kFrameBytes / 16; i += 64) fd[i] = fs[i];
          phase = PH_END;
          continue;
        }
      }
      BEGIN_EPISODIC_RESET(TO_B);  // VectorEnv auto-reset -> FrameStack.reset -> FireResetEnv.reset
    } else if (ctx == CTX_FIRE1) {  // FireResetEnv.reset :165-167
      if (d && mode == MODE_SNAPSHOT) emu.jam |= 0x4000;  // canned sequence deviated
      if (d) BEGIN_EPISODIC_RESET(TO_C);
      else BEGIN_SKIP(CTX_FIRE2, action_code(2));
    } else {  // CTX_FIRE2 :168-171: the obs of step(2) is returned even if a reset follows
      if (d) { no_render = 1; BEGIN_EPISODIC_RESET(TO_END); }
      else phase = PH_END;
    }
  }

  if (mode == MODE_SNAPSHOT) {
    // a done / closed episode inside the canned sequence would need the general path
    if (ep_closed || v.was_real_done) emu.jam |= 0x4000;
  } else if (lane == 0) {
    if (mode == MODE_STEP) {
      rewards[e] = (float)((out_total > 0) - (out_total < 0));  // ClipRewardEnv: np.sign
      dones[e] = out_done ? 1 : 0;
      ep_returns[e] = (float)ep_return;
      ep_lengths[e] = ep_closed ? ep_length : 0;
    }
    // bit 2: the env's step is not complete, no observation in this launch (frame_post skips the env)
    const int suspended = phase != PH_END;
    obs_flags[e] = (uint8_t)((suspended ? 4 : 0) | (did_reset ? 2 : 0) | (v.obs_single ? 1 : 0));
    if (emu.jam) atomicOr(jam_out, emu.jam);
    int* sc = (int*)(blob + kOffScalars);
    sc[S_SUSP] = suspended ? (0x400 | phase | (ctx << 2) | (cont << 4) | (skip_i << 6) | (no_render << 9)) : 0;
    sc[S_SUSP_TOTAL] = skip_total; sc[S_SUSP_ACT] = skip_act;
    sc[S_SUSP_ALE_J] = ale_j; sc[S_SUSP_NOOPS] = noops_left;
  }
  // the observation in this launch (parlhip_atari_vec_step_obs): the env's two waves convert its frame pair, half
  // the picture each, as soon as wave B has drawn it; the FrameStack counter and the MonitorEnv sums of the step
  // (what frame_post_kernel does on the side) are this wave's
  const StepFuse* fz = fuse_args();
  uint8_t* obs_out = mode == MODE_STEP ? fz->obs_out : nullptr;
  const bool do_obs = obs_out != nullptr && phase == PH_END;
#ifdef PARLHIP_ENV_REGIONS
  const unsigned long long k_tail = __builtin_readcyclecounter();   // last frame done -> end (store, wait for the picture, observation)
#endif
  // the frame flagged for wave B (fb_flags) was indeed the launch's last: its bands are being converted already
  const bool obs_banded = do_obs && !did_reset && !v.obs_single && emu.fb_flags == 1;
  emu.rq_ctl(Emu::LA_EXIT, (uint32_t)((do_obs ? 0x100 : 0) | (obs_banded ? 0x200 : 0) | (v.obs_single ? 1 : 0)), 0);
  emu.rq_flush();
  store_env(emu, v, blob, lane);
#ifdef PARLHIP_ENV_REGIONS
  const unsigned long long k_tail1 = __builtin_readcyclecounter();
  unsigned long long k_tail2 = k_tail1;
#endif
  if (obs_out) {
    if (lane == 0) {
      uint8_t* sn = fz->since_next;   // (null: elastic launches keep the FrameStack counters in elastic_post_kernel)
      if (sn) {
        const uint8_t* sp = fz->since_prev;
        const int p = sp ? sp[e] : 0;
        sn[e] = (phase != PH_END || did_reset) ? 0 : (uint8_t)(p + 1 > 3 ? 3 : p + 1);
      }
      double* acc = fz->ep_acc;
      if (acc && ep_closed && ep_length > 0) {
        atomicAdd(acc + 0, 1.0);
        atomicAdd(acc + 1, (double)(float)ep_return);
        atomicAdd(acc + 2, (double)ep_length);
      }
    }
    if (do_obs) {
      const int dim = fz->dim;
      if (obs_banded) {
        for (;;) {
          emu.obs_take();
          if (emu.obs_cur >= kObsBands) break;
          while (emu.obs_cur < emu.obs_end) {
            // as far as wave B has declared the bands final (all of them when it reaches LA_EXIT at the latest)
            const int rdy = (int)Emu::lds_ld(&emu.rq->obs_ready);
            const int n = ((rdy < emu.obs_end ? rdy : emu.obs_end) - emu.obs_cur) / kObsStep * kObsStep;
            if (n <= 0) { __builtin_amdgcn_s_sleep(2); continue; }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            obs_tail_dispatch(buf0, obs_out + (size_t)e * dim * dim, fz->tables, rom_lds + kMaxRomWords / 2, dim,
                              emu.obs_cur, emu.obs_cur + n, 0, wave);
            emu.obs_cur += n;
          }
        }
      } else {
        while (Emu::lds_ld(&emu.rq->fin) == 0u) __builtin_amdgcn_s_sleep(2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#ifdef PARLHIP_ENV_REGIONS
        k_tail2 = __builtin_readcyclecounter();
#endif
        obs_tail_dispatch(buf0, obs_out + (size_t)e * dim * dim, fz->tables, rom_lds + kMaxRomWords / 2, dim, 0,
                          kObsFirst, v.obs_single, wave);
      }
    }
  }
#ifdef PARLHIP_ENV_TIMING
  if (lane == 0 && e < 8192) g_env_t1[e] = wall_clock64();
#endif
#ifdef PARLHIP_ENV_REGIONS
  if (lane == 0 && e < 8192) {
    for (int i = 0; i < 4; ++i) { g_env_regions[e][i] = emu.rt[i]; g_env_regions[e][4 + i] = (unsigned long long)emu.rn[i]; }
    g_env_regions[e][8] = __builtin_readcyclecounter() - k0;
    // (packed: store | wait for the picture | observation, 21 bits each)
    { const unsigned long long k3 = __builtin_readcyclecounter();
      auto c21 = [](unsigned long long x) { return x > 0x1fffffull ? 0x1fffffull : x; };
      g_env_regions[e][16] = k_tail;
      g_env_regions[e][15] = c21(k_tail1 - k_tail) | (c21(k_tail2 - k_tail1) << 21) | (c21(k3 - k_tail2) << 42); }
  }
#endif
#ifdef PARLHIP_ENV_TRACEITER
  if (lane == 0 && e < 8192) { g_env_traceiter[e][0] = emu.ti_sum; g_env_traceiter[e][1] = emu.ti_cnt; g_env_traceiter[e][2] = emu.ti_other; }
#endif
}

// Row accounting of an elastic launch, before the emulator: who starts a row, who goes on, who waits.
// Rows are numbered per env from the start of the run (rows_done); row r lives at index r % rows_ring of
// the row tables; batch m = rows [m * batch_rows, (m + 1) * batch_rows).  An env may run ahead of the
// slowest one up to rows_limit (the caller raises it as batches complete).
__global__ void elastic_pre_kernel(const uint8_t* __restrict__ states, int E, int rows_limit, int rows_ring,
                                   int launch, int* __restrict__ rows_done, int* __restrict__ row_launch,
                                   int* __restrict__ row_slot, const int* __restrict__ cur_slot,
                                   uint8_t* __restrict__ ctl, uint8_t* __restrict__ obs_flags,
                                   int* __restrict__ ep_lengths) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int susp = ((const int*)(states + (size_t)e * kStateBytes + kOffScalars))[S_SUSP];
  if (susp) { ctl[e] = CTL_CONTINUE; return; }
  const int row = rows_done[e];
  if (row >= rows_limit) { ctl[e] = CTL_IDLE; obs_flags[e] = 4; ep_lengths[e] = 0; return; }
  ctl[e] = CTL_STEP;
  row_launch[(size_t)(row % rows_ring) * E + e] = launch;
  row_slot[(size_t)(row % rows_ring) * E + e] = cur_slot[e];  // the observation this row acts on
  rows_done[e] = row + 1;
}

// ... and after it: the row's reward / done (a plain step of 4 frames always fits the budget, so they
// are known in the launch that started the row); an env that has just started the last row of batch
// m reports it in finished[m & 1]
// Frame-stack bookkeeping of the elastic path lives here too: an env that completed its step in this
// launch gets the ring slot new_slot for its observation; link[new_slot][e] = the slot of its previous
// observation (launches it sat out leave gaps, so "the slot before" is not it), since = FrameStack's
// count of valid older frames (0 after a reset: four copies, atari_wrappers.py:290-294).
__global__ void elastic_post_kernel(int E, int rows_ring, int batch_rows, const int* __restrict__ rows_done,
                                    const uint8_t* __restrict__ ctl, const float* __restrict__ rewards,
                                    const uint8_t* __restrict__ dones, float* __restrict__ rewards_rows,
                                    uint8_t* __restrict__ dones_rows, int* __restrict__ finished,
                                    const uint8_t* __restrict__ obs_flags, int new_slot,
                                    int* __restrict__ cur_slot, int* __restrict__ link,
                                    uint8_t* __restrict__ since) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int fl = obs_flags[e];
  if (!(fl & 4)) {  // a new observation (frame_post writes it to ring[new_slot] next)
    const int old = cur_slot[e];
    const int p = since[(size_t)old * E + e];
    link[(size_t)new_slot * E + e] = old;
    since[(size_t)new_slot * E + e] = (fl & 2) ? 0 : (uint8_t)(p + 1 > 3 ? 3 : p + 1);
    cur_slot[e] = new_slot;
  }
  if (ctl[e] != CTL_STEP) return;
  const int rd = rows_done[e];
  const size_t at = (size_t)((rd - 1) % rows_ring) * E + e;
  rewards_rows[at] = rewards[e];
  dones_rows[at] = dones[e];
  if (rd % batch_rows == 0) atomicAdd(finished + ((rd / batch_rows - 1) & 1), 1);
}

}  // namespace atari
}  // namespace parlhip

using namespace parlhip;
using namespace parlhip::atari;

PARLHIP_EXPORT size_t parlhip_atari_state_bytes(void) { return kStateBytes; }
PARLHIP_EXPORT size_t parlhip_atari_frame_bytes(void) { return 2 * (size_t)kFrameBytes; }
PARLHIP_EXPORT size_t parlhip_atari_rom_table_bytes(uint32_t rom_size) { return (size_t)rom_size * 4; }
PARLHIP_EXPORT size_t parlhip_atari_reset_cache_bytes(void) { return kNumSnap * kSnapBytes; }

PARLHIP_EXPORT int parlhip_atari_rom_table_build(const uint8_t* rom_host, uint32_t rom_size,
                                                 uint32_t* table_host) {
  if (!rom_host || !table_host) return PARLHIP_EINVAL;
  if (rom_size != 2048 && rom_size != 4096) return PARLHIP_ENOSUP;  // unbanked 2K/4K carts
  build_rom_words(rom_host, rom_size, table_host);
  // tag the table (free top bits of word 0) when this is a cartridge the library carries
  // natively translated code for
  uint32_t crc = 0xffffffffu;
  for (uint32_t i = 0; i < rom_size; ++i) {
    crc ^= rom_host[i];
    for (int k = 0; k < 8; ++k) crc = (crc >> 1) ^ (0xedb88320u & (0u - (crc & 1u)));
  }
  crc = ~crc;
  uint32_t tag = 0;
  if (NativeCart<GAME_PONG>::present && crc == NativeCart<GAME_PONG>::rom_crc32) tag = GAME_PONG;
  if (NativeCart<GAME_BREAKOUT>::present && crc == NativeCart<GAME_BREAKOUT>::rom_crc32) tag = GAME_BREAKOUT;
  table_host[0] = (table_host[0] & 0x0fffffffu) | (tag << 28);
  return PARLHIP_OK;
}

PARLHIP_EXPORT uint32_t parlhip_atari_native_cart(int game) {
  if (game == GAME_PONG && NativeCart<GAME_PONG>::present) return NativeCart<GAME_PONG>::rom_crc32;
  if (game == GAME_BREAKOUT && NativeCart<GAME_BREAKOUT>::present) return NativeCart<GAME_BREAKOUT>::rom_crc32;
  return 0;
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

static int launch_env(int mode, void* states, const uint32_t* romw, uint32_t rom_size, int game,
                      const int64_t* actions, uint8_t* frames, float* rewards, uint8_t* dones,
                      uint8_t* obs_flags, float* ep_returns, int32_t* ep_lengths, int E, uint64_t seed,
                      uint64_t env_id0, int64_t max_steps, void* snap, int32_t* jam, hipStream_t s,
                      int budget = 0, const uint8_t* ctl = nullptr, const StepFuse* fuse = nullptr) {
  const StepFuse fz = fuse ? *fuse : StepFuse{};
  EnvParams prm{game, (int)rom_size, E, mode, seed, env_id0, (long long)max_steps, budget};
  const dim3 grid(ceil_div(E, kEnvsPerBlock)), block(128 * kEnvsPerBlock);
#define PARLHIP_LAUNCH_ENV(G)                                                                           \
  atari_env_kernel<G><<<grid, block, 0, s>>>(fz, (uint8_t*)states, romw, prm, (const long long*)actions, \
                                             frames, rewards, dones, obs_flags, ep_returns, ep_lengths, \
                                             (uint8_t*)snap, jam, ctl)
  if (game == GAME_PONG) PARLHIP_LAUNCH_ENV(GAME_PONG);
  else PARLHIP_LAUNCH_ENV(GAME_BREAKOUT);
#undef PARLHIP_LAUNCH_ENV
  return check_launch();
}

PARLHIP_EXPORT int parlhip_atari_reset_cache_build(const uint32_t* rom_table_dev, uint32_t rom_size,
                                                   int game, int64_t max_episode_steps, void* cache_dev,
                                                   int32_t* jam_flag_dev, parlhip_stream_t stream) {
  int rc = check_env_args((void*)1, rom_table_dev, rom_size, game, 1);
  if (rc) return rc;
  if (!cache_dev || !jam_flag_dev) return PARLHIP_EINVAL;
  return launch_env(MODE_SNAPSHOT, nullptr, rom_table_dev, rom_size, game, nullptr, nullptr, nullptr,
                    nullptr, nullptr, nullptr, nullptr, kNumSnap, 0, 0, max_episode_steps, cache_dev,
                    jam_flag_dev, (hipStream_t)stream);
}

PARLHIP_EXPORT int parlhip_atari_vec_reset(void* states, const uint32_t* rom_table_dev, uint32_t rom_size,
                                           int game, uint8_t* frames, uint8_t* obs_flags, int E,
                                           uint64_t seed, uint64_t env_id0, int64_t max_episode_steps,
                                           int32_t* jam_flag_dev, parlhip_stream_t stream) {
  int rc = check_env_args(states, rom_table_dev, rom_size, game, E);
  if (rc) return rc;
  if (E == 0) return PARLHIP_OK;
  if (!frames || !obs_flags || !jam_flag_dev) return PARLHIP_EINVAL;
  return launch_env(MODE_RESET, states, rom_table_dev, rom_size, game, nullptr, frames, nullptr, nullptr,
                    obs_flags, nullptr, nullptr, E, seed, env_id0, max_episode_steps, nullptr, jam_flag_dev,
                    (hipStream_t)stream);
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
  return launch_env(MODE_STEP, states, rom_table_dev, rom_size, game, actions, frames, rewards, dones,
                    obs_flags, ep_returns, ep_lengths, E, seed, env_id0, max_episode_steps,
                    (void*)reset_cache_dev, jam_flag_dev, (hipStream_t)stream);
}

PARLHIP_EXPORT int parlhip_atari_vec_step_obs(void* states, const uint32_t* rom_table_dev, uint32_t rom_size, int game,
                                              const int64_t* actions, uint8_t* frames, float* rewards,
                                              uint8_t* dones, uint8_t* obs_flags, float* ep_returns,
                                              int32_t* ep_lengths, int E, uint64_t seed, uint64_t env_id0,
                                              int64_t max_episode_steps, const void* reset_cache_dev,
                                              int32_t* jam_flag_dev, uint8_t* obs_out, int dim,
                                              const void* tables_dev, const uint8_t* since_prev,
                                              uint8_t* since_next, double* ep_acc3, parlhip_stream_t stream) {
  int rc = check_env_args(states, rom_table_dev, rom_size, game, E);
  if (rc) return rc;
  if (!obs_tail_supports(dim, (int)rom_size)) return PARLHIP_ENOSUP;  // the caller launches parlhip_frame_post_step_u8 instead
  if (E == 0) return PARLHIP_OK;
  if (!actions || !frames || !rewards || !dones || !obs_flags || !ep_returns || !ep_lengths || !jam_flag_dev ||
      !obs_out || !tables_dev || !since_next)
    return PARLHIP_EINVAL;
  if (reinterpret_cast<uintptr_t>(frames) & 15) return PARLHIP_EINVAL;   // the tail reads the frame pair as uint4s
  StepFuse fz{};
  fz.obs_out = obs_out; fz.tables = (const uint8_t*)tables_dev; fz.since_prev = since_prev; fz.since_next = since_next;
  fz.ep_acc = ep_acc3; fz.dim = dim;
  return launch_env(MODE_STEP, states, rom_table_dev, rom_size, game, actions, frames, rewards, dones,
                    obs_flags, ep_returns, ep_lengths, E, seed, env_id0, max_episode_steps,
                    (void*)reset_cache_dev, jam_flag_dev, (hipStream_t)stream, 0, nullptr, &fz);
}

PARLHIP_EXPORT int parlhip_atari_vec_step_policy_obs(
    void* states, const uint32_t* rom_table_dev, uint32_t rom_size, int game, uint8_t* frames, float* rewards,
    uint8_t* dones, uint8_t* obs_flags, float* ep_returns, int32_t* ep_lengths, int E, uint64_t seed,
    uint64_t env_id0, int64_t max_episode_steps, const void* reset_cache_dev, int32_t* jam_flag_dev,
    uint8_t* obs_out, int dim, const void* tables_dev, const uint8_t* since_prev, uint8_t* since_next,
    double* ep_acc3, const float* hidden, const float* w_policy, const float* b_policy, float* logits_out,
    int64_t* actions_out, int hidden_units, int A, uint64_t sample_seed, const uint64_t* offset_base,
    uint64_t offset, uint64_t row0, parlhip_stream_t stream) {
  int rc = check_env_args(states, rom_table_dev, rom_size, game, E);
  if (rc) return rc;
  if (!obs_tail_supports(dim, (int)rom_size)) return PARLHIP_ENOSUP;
  if (hidden_units != 256 || A < 1 || A > 6) return PARLHIP_ENOSUP;
  if (A != parlhip_atari_num_actions(game)) return PARLHIP_EINVAL;
  if (E == 0) return PARLHIP_OK;
  if (!frames || !rewards || !dones || !obs_flags || !ep_returns || !ep_lengths || !jam_flag_dev || !obs_out ||
      !tables_dev || !since_next || !hidden || !w_policy || !b_policy || !logits_out || !actions_out)
    return PARLHIP_EINVAL;
  if ((reinterpret_cast<uintptr_t>(frames) | reinterpret_cast<uintptr_t>(hidden) |
       reinterpret_cast<uintptr_t>(w_policy)) & 15)
    return PARLHIP_EINVAL;
  StepFuse fz{};
  fz.obs_out = obs_out; fz.tables = (const uint8_t*)tables_dev; fz.since_prev = since_prev; fz.since_next = since_next;
  fz.ep_acc = ep_acc3; fz.dim = dim; fz.A = A;
  fz.hidden = hidden; fz.w_pi = w_policy; fz.b_pi = b_policy; fz.logits_out = logits_out;
  fz.actions_out = (long long*)actions_out; fz.offset_base = (const unsigned long long*)offset_base;
  fz.sample_seed = sample_seed; fz.offset = offset; fz.row0 = row0;
  return launch_env(MODE_STEP, states, rom_table_dev, rom_size, game, (const int64_t*)actions_out, frames, rewards,
                    dones, obs_flags, ep_returns, ep_lengths, E, seed, env_id0, max_episode_steps,
                    (void*)reset_cache_dev, jam_flag_dev, (hipStream_t)stream, 0, nullptr, &fz);
}

static int vec_step_elastic(const StepFuse* fuse, void* states, const uint32_t* rom_table_dev, uint32_t rom_size,
                                                  int game, const int64_t* actions, uint8_t* frames, float* rewards,
                                                  uint8_t* dones, uint8_t* obs_flags, float* ep_returns,
                                                  int32_t* ep_lengths, int E, uint64_t seed, uint64_t env_id0,
                                                  int64_t max_episode_steps, const void* reset_cache_dev,
                                                  int32_t* jam_flag_dev, int frame_budget, int launch,
                                                  int rows_limit, int rows_ring, int batch_rows,
                                                  int32_t* rows_done, int32_t* row_launch, int32_t* row_slot,
                                                  uint8_t* ctl, int32_t* finished, float* rewards_rows,
                                                  uint8_t* dones_rows, int new_slot, int32_t* cur_slot,
                                                  int32_t* link, uint8_t* since, parlhip_stream_t stream) {
  int rc = check_env_args(states, rom_table_dev, rom_size, game, E);
  if (rc) return rc;
  if (E == 0) return PARLHIP_OK;
  if (!actions || !frames || !rewards || !dones || !obs_flags || !ep_returns || !ep_lengths || !jam_flag_dev ||
      !rows_done || !row_launch || !row_slot || !ctl || !finished || !rewards_rows || !dones_rows || !cur_slot ||
      !link || !since || new_slot < 0)
    return PARLHIP_EINVAL;
  // a plain step (4 frames) must fit the budget; the row tables hold two batches
  if (frame_budget < 4 || launch < 0 || batch_rows < 1 || rows_ring < 2 * batch_rows || rows_ring % batch_rows ||
      rows_limit < 0)
    return PARLHIP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  elastic_pre_kernel<<<ceil_div(E, 256), 256, 0, s>>>((const uint8_t*)states, E, rows_limit, rows_ring, launch,
                                                      rows_done, row_launch, row_slot, cur_slot, ctl, obs_flags,
                                                      ep_lengths);
  rc = launch_env(MODE_STEP, states, rom_table_dev, rom_size, game, actions, frames, rewards, dones, obs_flags,
                  ep_returns, ep_lengths, E, seed, env_id0, max_episode_steps, (void*)reset_cache_dev,
                  jam_flag_dev, s, frame_budget, ctl, fuse);
  if (rc) return rc;
  elastic_post_kernel<<<ceil_div(E, 256), 256, 0, s>>>(E, rows_ring, batch_rows, rows_done, ctl, rewards, dones,
                                                       rewards_rows, dones_rows, finished, obs_flags, new_slot,
                                                       cur_slot, link, since);
  return check_launch();
}

PARLHIP_EXPORT int parlhip_atari_vec_step_elastic(void* states, const uint32_t* rom_table_dev, uint32_t rom_size,
                                                  int game, const int64_t* actions, uint8_t* frames, float* rewards,
                                                  uint8_t* dones, uint8_t* obs_flags, float* ep_returns,
                                                  int32_t* ep_lengths, int E, uint64_t seed, uint64_t env_id0,
                                                  int64_t max_episode_steps, const void* reset_cache_dev,
                                                  int32_t* jam_flag_dev, int frame_budget, int launch,
                                                  int rows_limit, int rows_ring, int batch_rows,
                                                  int32_t* rows_done, int32_t* row_launch, int32_t* row_slot,
                                                  uint8_t* ctl, int32_t* finished, float* rewards_rows,
                                                  uint8_t* dones_rows, int new_slot, int32_t* cur_slot,
                                                  int32_t* link, uint8_t* since, parlhip_stream_t stream) {
  return vec_step_elastic(nullptr, states, rom_table_dev, rom_size, game, actions, frames, rewards, dones, obs_flags,
                          ep_returns, ep_lengths, E, seed, env_id0, max_episode_steps, reset_cache_dev, jam_flag_dev,
                          frame_budget, launch, rows_limit, rows_ring, batch_rows, rows_done, row_launch, row_slot, ctl,
                          finished, rewards_rows, dones_rows, new_slot, cur_slot, link, since, stream);
}

PARLHIP_EXPORT int parlhip_atari_vec_step_elastic_obs(void* states, const uint32_t* rom_table_dev, uint32_t rom_size,
                                                      int game, const int64_t* actions, uint8_t* frames, float* rewards,
                                                      uint8_t* dones, uint8_t* obs_flags, float* ep_returns,
                                                      int32_t* ep_lengths, int E, uint64_t seed, uint64_t env_id0,
                                                      int64_t max_episode_steps, const void* reset_cache_dev,
                                                      int32_t* jam_flag_dev, int frame_budget, int launch,
                                                      int rows_limit, int rows_ring, int batch_rows,
                                                      int32_t* rows_done, int32_t* row_launch, int32_t* row_slot,
                                                      uint8_t* ctl, int32_t* finished, float* rewards_rows,
                                                      uint8_t* dones_rows, int new_slot, int32_t* cur_slot,
                                                      int32_t* link, uint8_t* since, uint8_t* obs_out, int dim,
                                                      const void* tables_dev, parlhip_stream_t stream) {
  if (!obs_tail_supports(dim, (int)rom_size)) return PARLHIP_ENOSUP;
  if (E > 0 && (!obs_out || !tables_dev)) return PARLHIP_EINVAL;
  if (reinterpret_cast<uintptr_t>(frames) & 15) return PARLHIP_EINVAL;
  StepFuse fz{};
  fz.obs_out = obs_out; fz.tables = (const uint8_t*)tables_dev; fz.dim = dim;   // since / MonitorEnv sums: the elastic path's own kernels
  return vec_step_elastic(&fz, states, rom_table_dev, rom_size, game, actions, frames, rewards, dones, obs_flags,
                          ep_returns, ep_lengths, E, seed, env_id0, max_episode_steps, reset_cache_dev, jam_flag_dev,
                          frame_budget, launch, rows_limit, rows_ring, batch_rows, rows_done, row_launch, row_slot, ctl,
                          finished, rewards_rows, dones_rows, new_slot, cur_slot, link, since, stream);
}

#ifdef PARLHIP_ENV_REGIONS
PARLHIP_EXPORT int parlhip_debug_env_regions(unsigned long long* host, int n) {
  if (n > 8192) return PARLHIP_EINVAL;
#ifdef PARLHIP_ENV_ENTRYHIST
  if (n == -8192) {
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(parlhip::atari::g_env_entryhist), sizeof(unsigned long long) * 8192 * 3) != hipSuccess) return PARLHIP_EINVAL;
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(parlhip::atari::g_env_entryhist)) != hipSuccess) return PARLHIP_EINVAL;
    return hipMemset(p, 0, sizeof(unsigned long long) * 8192 * 3) == hipSuccess ? PARLHIP_OK : PARLHIP_EINVAL;
  }
#endif
#ifdef PARLHIP_ENV_TRACEITER
  if (n < 0) return hipMemcpyFromSymbol(host, HIP_SYMBOL(parlhip::atari::g_env_traceiter), (size_t)(-n) * 128) == hipSuccess ? PARLHIP_OK : PARLHIP_EINVAL;
#endif
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(parlhip::atari::g_env_regions), (size_t)n * 192) != hipSuccess) return PARLHIP_EINVAL;
  return PARLHIP_OK;
}
#endif

#ifdef PARLHIP_ENV_TIMING
PARLHIP_EXPORT int parlhip_debug_env_clocks(unsigned long long* t0_host, unsigned long long* t1_host, int n) {
  if (n > 8192) return PARLHIP_EINVAL;
  if (hipMemcpyFromSymbol(t0_host, HIP_SYMBOL(parlhip::atari::g_env_t0), (size_t)n * 8) != hipSuccess) return PARLHIP_EINVAL;
  if (hipMemcpyFromSymbol(t1_host, HIP_SYMBOL(parlhip::atari::g_env_t1), (size_t)n * 8) != hipSuccess) return PARLHIP_EINVAL;
  return PARLHIP_OK;
}
#endif
