// scan_kernels.hip — V-trace / GAE / discounted-sum reverse scans for gfx950 (MI355X).
//
// Reference arithmetic (paths relative to the PARL tree):
//   parl/algorithms/paddle/impala/vtrace.py:36-139   from_importance_weights
//   parl/algorithms/paddle/impala/impala.py:59,119-132,167-194   discounts, _log_prob, slicing
//   parl/utils/rl_utils.py:21-51 + examples/A2C/actor.py:73-85   calc_gae / segments
//   examples/PPO/storage.py:45-64                                 RolloutStorage.compute_returns
//
// All of these are HBM-bound reverse linear recurrences over [T,B] float32 slabs.  Two
// mappings are used:
//   * lane-per-sequence (time-major input, B contiguous): every load of a wave is one
//     coalesced 256 B (VEC=1) or 1 KiB (VEC=4, float4) segment; the carry lives in registers
//     and the loads of U consecutive time steps are issued before the dependent FMA chain so
//     U*5 wide loads per lane are in flight.
//   * wave-per-sequence (env-major input, T contiguous — the reference's flat [B*T] batch):
//     lane l owns time steps [l*K, l*K+K), loads are coalesced along T, and the affine
//     recurrence acc_t = d_t + a_t*acc_{t+1} is solved with a 6-step wavefront-shuffle suffix
//     scan over (a, d) pairs; neighbours' V_{t+1} / vs_{t+1} come from one more shuffle.
#include <cstdlib>
#include "common.hpp"
#include <math.h>

namespace parlhip {

// ----------------------------------------------------------------------------------------
// small vector helpers (VEC = 1 or 4 sequences per lane)
// ----------------------------------------------------------------------------------------
template <int VEC> struct Vec;
template <> struct Vec<1> {
  float v[1];
  __device__ static Vec load(const float* p) { Vec r; r.v[0] = *p; return r; }
  __device__ void store(float* p) const { *p = v[0]; }
  __device__ void store_nt(float* p) const { __builtin_nontemporal_store(v[0], p); }
};
template <> struct Vec<2> {
  float v[2];
  __device__ static Vec load(const float* p) {
    float2 q = *reinterpret_cast<const float2*>(p);
    Vec r; r.v[0] = q.x; r.v[1] = q.y; return r;
  }
  __device__ void store(float* p) const { *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]); }
  __device__ void store_nt(float* p) const {
    __builtin_nontemporal_store(v[0], p); __builtin_nontemporal_store(v[1], p + 1);
  }
};
template <> struct Vec<4> {
  float v[4];
  __device__ static Vec load(const float* p) {
    float4 q = *reinterpret_cast<const float4*>(p);
    Vec r; r.v[0] = q.x; r.v[1] = q.y; r.v[2] = q.z; r.v[3] = q.w; return r;
  }
  __device__ void store(float* p) const {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
  __device__ void store_nt(float* p) const {
    __builtin_nontemporal_store(v[0], p); __builtin_nontemporal_store(v[1], p + 1);
    __builtin_nontemporal_store(v[2], p + 2); __builtin_nontemporal_store(v[3], p + 3);
  }
};

template <int VEC> struct U8Vec;
template <> struct U8Vec<1> {
  uint8_t v[1];
  __device__ static U8Vec load(const uint8_t* p) { U8Vec r; r.v[0] = *p; return r; }
};
template <> struct U8Vec<4> {
  uint8_t v[4];
  __device__ static U8Vec load(const uint8_t* p) {
    uint32_t q = *reinterpret_cast<const uint32_t*>(p);
    U8Vec r; r.v[0] = q & 0xff; r.v[1] = (q >> 8) & 0xff; r.v[2] = (q >> 16) & 0xff;
    r.v[3] = (q >> 24) & 0xff; return r;
  }
};

__device__ __forceinline__ float clip_max(float x, float thr) {
  // NaN threshold == the reference's `None`: no clipping (vtrace.py:102-105)
  return (thr != thr) ? x : fminf(x, thr);
}

// ----------------------------------------------------------------------------------------
// V-trace, lane-per-sequence, time-major, from log-probs.  28 B per (t,b) element.
// ----------------------------------------------------------------------------------------
template <int VEC, int U, bool NT = false>
__global__ __launch_bounds__(256) void vtrace_tm_kernel(
    const float* __restrict__ blp, const float* __restrict__ tlp,
    const float* __restrict__ disc, const float* __restrict__ rew,
    const float* __restrict__ val, const float* __restrict__ boot,
    float* __restrict__ vs, float* __restrict__ pg, int T, int B, float clip_rho,
    float clip_pg) {
  const int64_t b0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
  if (b0 >= B) return;
  float acc[VEC], vs_next[VEC], v_next[VEC];
  {
    Vec<VEC> bv = Vec<VEC>::load(boot + b0);
#pragma unroll
    for (int j = 0; j < VEC; ++j) { acc[j] = 0.f; vs_next[j] = bv.v[j]; v_next[j] = bv.v[j]; }
  }
  int t = T - 1;
  // main loop: U time steps per iteration, all loads issued before the dependent chain
  for (; t - (U - 1) >= 0; t -= U) {
    Vec<VEC> l_b[U], l_t[U], l_d[U], l_r[U], l_v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = (int64_t)(t - u) * B + b0;
      l_b[u] = Vec<VEC>::load(blp + i);
      l_t[u] = Vec<VEC>::load(tlp + i);
      l_d[u] = Vec<VEC>::load(disc + i);
      l_r[u] = Vec<VEC>::load(rew + i);
      l_v[u] = Vec<VEC>::load(val + i);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = (int64_t)(t - u) * B + b0;
      Vec<VEC> o_vs, o_pg;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float rho = expf(l_t[u].v[j] - l_b[u].v[j]);
        const float crho = clip_max(rho, clip_rho);
        const float c = fminf(rho, 1.0f);
        const float d = l_d[u].v[j], v = l_v[u].v[j], r = l_r[u].v[j];
        const float delta = crho * (r + d * v_next[j] - v);
        acc[j] = delta + d * c * acc[j];
        const float vst = acc[j] + v;
        o_pg.v[j] = clip_max(rho, clip_pg) * (r + d * vs_next[j] - v);
        o_vs.v[j] = vst;
        vs_next[j] = vst;
        v_next[j] = v;
      }
      if (NT) { o_vs.store_nt(vs + i); o_pg.store_nt(pg + i); }
      else { o_vs.store(vs + i); o_pg.store(pg + i); }
    }
  }
  for (; t >= 0; --t) {
    const int64_t i = (int64_t)t * B + b0;
    Vec<VEC> xb = Vec<VEC>::load(blp + i), xt = Vec<VEC>::load(tlp + i),
             xd = Vec<VEC>::load(disc + i), xr = Vec<VEC>::load(rew + i),
             xv = Vec<VEC>::load(val + i), o_vs, o_pg;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float rho = expf(xt.v[j] - xb.v[j]);
      const float crho = clip_max(rho, clip_rho);
      const float c = fminf(rho, 1.0f);
      const float d = xd.v[j], v = xv.v[j], r = xr.v[j];
      const float delta = crho * (r + d * v_next[j] - v);
      acc[j] = delta + d * c * acc[j];
      const float vst = acc[j] + v;
      o_pg.v[j] = clip_max(rho, clip_pg) * (r + d * vs_next[j] - v);
      o_vs.v[j] = vst;
      vs_next[j] = vst;
      v_next[j] = v;
    }
    if (NT) { o_vs.store_nt(vs + i); o_pg.store_nt(pg + i); }
    else { o_vs.store(vs + i); o_pg.store(pg + i); }
  }
}

// ----------------------------------------------------------------------------------------
// log-softmax gather (IMPALA._log_prob, impala.py:119-132)
// ----------------------------------------------------------------------------------------
template <int A_CT>
__device__ __forceinline__ float log_prob_row(const float* __restrict__ row, int A, int a) {
  const int n = A_CT > 0 ? A_CT : A;
  if (A_CT > 0) {
    float x[A_CT > 0 ? A_CT : 1];
#pragma unroll
    for (int k = 0; k < A_CT; ++k) x[k] = row[k];
    float m = x[0];
#pragma unroll
    for (int k = 1; k < A_CT; ++k) m = fmaxf(m, x[k]);
    float s = 0.f, xa = x[0];
#pragma unroll
    for (int k = 0; k < A_CT; ++k) {
      s += expf(x[k] - m);
      xa = (k == a) ? x[k] : xa;
    }
    return (xa - m) - logf(s);
  } else {
    float m = row[0];
    for (int k = 1; k < n; ++k) m = fmaxf(m, row[k]);
    float s = 0.f;
    for (int k = 0; k < n; ++k) s += expf(row[k] - m);
    return (row[a] - m) - logf(s);
  }
}

// V-trace fused from logits, lane-per-sequence, time-major [T,B,A].
template <int A_CT>
__global__ __launch_bounds__(256) void vtrace_logits_tm_kernel(
    const float* __restrict__ blog, const float* __restrict__ tlog,
    const int64_t* __restrict__ actions, const float* __restrict__ rew,
    const uint8_t* __restrict__ dones, const float* __restrict__ val,
    float* __restrict__ vs, float* __restrict__ pg, float* __restrict__ tlp_out,
    float* __restrict__ blp_out, int T, int B, int A, float gamma, float clip_rho,
    float clip_pg, int* __restrict__ err) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float bootstrap = val[(int64_t)(T - 1) * B + b];
  float acc = 0.f, vs_next = bootstrap, v_next = bootstrap;
#pragma unroll 2
  for (int t = T - 2; t >= 0; --t) {
    const int64_t i = (int64_t)t * B + b;
    int a = (int)actions[i];
    if (a < 0 || a >= A) { *err = 1; a = 0; }
    const float tl = log_prob_row<A_CT>(tlog + i * A, A, a);
    const float bl = log_prob_row<A_CT>(blog + i * A, A, a);
    const float d = dones[i] ? 0.f : gamma;
    const float rho = expf(tl - bl);
    const float crho = clip_max(rho, clip_rho);
    const float c = fminf(rho, 1.0f);
    const float v = val[i], r = rew[i];
    const float delta = crho * (r + d * v_next - v);
    acc = delta + d * c * acc;
    const float vst = acc + v;
    pg[i] = clip_max(rho, clip_pg) * (r + d * vs_next - v);
    vs[i] = vst;
    if (tlp_out) tlp_out[i] = tl;
    if (blp_out) blp_out[i] = bl;
    vs_next = vst;
    v_next = v;
  }
}

// ----------------------------------------------------------------------------------------
// Wave-per-sequence V-trace core.  Lane l owns time steps t = l*K + k, k < K (T' <= 64*K); the
// affine recurrence acc_t = delta_t + (disc_t * c_t) * acc_{t+1} is solved with a 6-step
// wavefront-shuffle suffix scan over (a, d) pairs, so one sequence costs ONE round of loads
// plus O(K + log 64) dependent steps instead of T' dependent steps.  Used (a) for the
// reference's env-major flat batch, where T is the contiguous axis, and (b) for time-major
// input when B is too small for lane-per-sequence to fill the chip (B=1024: 16 waves).
// ----------------------------------------------------------------------------------------
template <int K>
__device__ __forceinline__ void vtrace_wave_core(const float (&rho)[K], const float (&dsc)[K],
                                                 const float (&v)[K], const float (&r)[K],
                                                 const bool (&valid)[K], int lane, int Tm,
                                                 float bootstrap, float clip_rho, float clip_pg,
                                                 float (&vst)[K], float (&pgv)[K]) {
  // V_{t+1}: next step in-lane, or the first value of the next lane, or the bootstrap.
  const float v_first_next_lane = __shfl_down(v[0], 1, 64);
  float v_next[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int t = lane * K + k;
    float nv = (k + 1 < K) ? v[(k + 1 < K) ? k + 1 : k] : v_first_next_lane;
    if (t + 1 >= Tm) nv = bootstrap;
    v_next[k] = nv;
  }
  // per-step affine maps acc_t = d_t + a_t * acc_{t+1}; invalid steps are identity.
  float a_[K], d_[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float crho = clip_max(rho[k], clip_rho);
    const float c = fminf(rho[k], 1.0f);
    d_[k] = valid[k] ? crho * (r[k] + dsc[k] * v_next[k] - v[k]) : 0.f;
    a_[k] = valid[k] ? dsc[k] * c : 1.f;
  }
  // lane-local composition over [l*K, l*K+K): (LA, LD)
  float LA = 1.f, LD = 0.f;
#pragma unroll
  for (int k = K - 1; k >= 0; --k) {
    LD = d_[k] + a_[k] * LD;
    LA = a_[k] * LA;
  }
  // inclusive suffix scan across lanes (Hillis–Steele, 6 shuffle steps)
  float SA = LA, SD = LD;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float oa = __shfl_down(SA, off, 64);
    const float od = __shfl_down(SD, off, 64);
    if (lane + off < 64) {
      SD = SD + SA * od;
      SA = SA * oa;
    }
  }
  // carry entering this lane's block = suffix result of lane+1 (0 for the last lane)
  float carry = __shfl_down(SD, 1, 64);
  if (lane == 63) carry = 0.f;
#pragma unroll
  for (int k = K - 1; k >= 0; --k) {
    carry = d_[k] + a_[k] * carry;
    vst[k] = carry + v[k];
  }
  const float vs_first_next_lane = __shfl_down(vst[0], 1, 64);
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int t = lane * K + k;
    float nvs = (k + 1 < K) ? vst[(k + 1 < K) ? k + 1 : k] : vs_first_next_lane;
    if (t + 1 >= Tm) nvs = bootstrap;
    pgv[k] = clip_max(rho[k], clip_pg) * (r[k] + dsc[k] * nvs - v[k]);
  }
}

// Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  In the time-major
// wave-per-sequence kernels neighbouring sequences share cache lines (sequence b reads 4 or 4A
// bytes at stride B per step), so workgroup i and i+1 on different XCDs make every line cross the
// fabric up to 8 times (PMC: 2.9x the algorithmic bytes at T=50, B=1024, A=6).  This maps XCD x
// onto the contiguous chunk x of the workgroup range instead (bijective for any grid size).
__device__ __forceinline__ int xcd_chunk_block(int bid, int nb) {
  const int x = bid & (kNumXCD - 1), j = bid >> 3;
  const int q = nb >> 3, rem = nb & (kNumXCD - 1);
  return x * q + (x < rem ? x : rem) + j;
}

// V-trace fused from logits, wave-per-sequence.  TM = false: env-major [B,T,A] (the reference's
// flat batch, impala.py:167-175); TM = true: time-major [T,B,A] with small B.
template <int A_CT, int K, bool TM>
__global__ __launch_bounds__(256) void vtrace_logits_wave_kernel(
    const float* __restrict__ blog, const float* __restrict__ tlog,
    const int64_t* __restrict__ actions, const float* __restrict__ rew,
    const uint8_t* __restrict__ dones, const float* __restrict__ val,
    float* __restrict__ vs, float* __restrict__ pg, float* __restrict__ tlp_out,
    float* __restrict__ blp_out, int T, int B, int A, float gamma, float clip_rho,
    float clip_pg, int* __restrict__ err) {
  const int lane = threadIdx.x & 63;
  const int blk = TM ? xcd_chunk_block(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int64_t b = ((int64_t)blk * blockDim.x + threadIdx.x) >> 6;
  if (b >= B) return;  // whole wave exits together
  const int Tm = T - 1;
  // element (t, b): inputs hold T steps, outputs T-1 steps, same major order
  const int64_t in_t = TM ? B : 1, in_b = TM ? 1 : T, out_b = TM ? 1 : Tm;
  const float bootstrap = val[(int64_t)Tm * in_t + b * in_b];

  float rho[K], dsc[K], v[K], r[K], tl[K], bl[K], vst[K], pgv[K];
  bool valid[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int t = lane * K + k;
    valid[k] = t < Tm;
    rho[k] = 1.f; dsc[k] = 0.f; v[k] = 0.f; r[k] = 0.f; tl[k] = 0.f; bl[k] = 0.f;
    if (valid[k]) {
      const int64_t i = (int64_t)t * in_t + b * in_b;
      int a = (int)actions[i];
      if (a < 0 || a >= A) { *err = 1; a = 0; }
      tl[k] = log_prob_row<A_CT>(tlog + i * A, A, a);
      bl[k] = log_prob_row<A_CT>(blog + i * A, A, a);
      dsc[k] = dones[i] ? 0.f : gamma;
      rho[k] = expf(tl[k] - bl[k]);
      v[k] = val[i];
      r[k] = rew[i];
    }
  }
  vtrace_wave_core<K>(rho, dsc, v, r, valid, lane, Tm, bootstrap, clip_rho, clip_pg, vst, pgv);
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int t = lane * K + k;
    if (!valid[k]) continue;
    const int64_t o = (int64_t)t * in_t + b * out_b;
    pg[o] = pgv[k];
    vs[o] = vst[k];
    if (tlp_out) tlp_out[o] = tl[k];
    if (blp_out) blp_out[o] = bl[k];
  }
}

// ----------------------------------------------------------------------------------------
// IMPALA learner loss in one pass: the fused V-trace above plus everything
// IMPALA.learn computes around it — log-softmax of both policies, the action log-prob gather,
// Categorical entropy and KL (impala.py:119-165), pi / vf / entropy sums (impala.py:67-79) — and
// the gradient of   total = pi_loss + vf_coeff * vf_loss + ent_coeff * entropy   with respect to
// the target logits and the values (vs / pg_advantages carry no gradient: vtrace.py:36):
//   d total / d logit_j = -pg_adv * (1[j == a] - p_j) + ent_coeff * (-p_j * (log p_j + H))
//   d total / d V       = vf_coeff * (V - vs)
// for the T-1 transitions, zero for the bootstrap row.  ~20 eager launches of the autograd graph
// become this kernel plus two multiplies in backward.  sums (float64, caller-zeroed):
// [0] pi_loss [1] vf_loss [2] entropy [3] sum over ALL T rows of KL(target || behaviour).
// ----------------------------------------------------------------------------------------
template <int A_CT>
__device__ __forceinline__ void log_softmax_row(const float* __restrict__ row, float (&lp)[A_CT]) {
  float x[A_CT];
#pragma unroll
  for (int j = 0; j < A_CT; ++j) x[j] = row[j];
  float m = x[0];
#pragma unroll
  for (int j = 1; j < A_CT; ++j) m = fmaxf(m, x[j]);
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < A_CT; ++j) sum += expf(x[j] - m);
  const float lse = logf(sum);
#pragma unroll
  for (int j = 0; j < A_CT; ++j) lp[j] = (x[j] - m) - lse;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int A_CT, int K, bool TM>
__global__ __launch_bounds__(256) void impala_loss_wave_kernel(
    const float* __restrict__ blog, const float* __restrict__ tlog,
    const int64_t* __restrict__ actions, const float* __restrict__ rew,
    const uint8_t* __restrict__ dones, const float* __restrict__ val,
    float* __restrict__ vs, float* __restrict__ pg, float* __restrict__ glog,
    float* __restrict__ gval, double* __restrict__ sums, int T, int B, float gamma,
    float clip_rho, float clip_pg, float vf_coeff, float ent_coeff, int* __restrict__ err) {
  const int lane = threadIdx.x & 63;
  const int blk = TM ? xcd_chunk_block(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int64_t b_raw = ((int64_t)blk * blockDim.x + threadIdx.x) >> 6;
  // a wave past the last sequence recomputes sequence B-1 with its stores and sums masked, so that
  // the whole workgroup reaches the reduction barrier below
  const bool live = b_raw < B;
  const int64_t b = live ? b_raw : (int64_t)B - 1;
  const int Tm = T - 1;
  const int64_t in_t = TM ? B : 1, in_b = TM ? 1 : T, out_b = TM ? 1 : Tm;
  const float bootstrap = val[(int64_t)Tm * in_t + b * in_b];

  float rho[K], dsc[K], v[K], r[K], vst[K], pgv[K], tlp[K], H[K];
  float p[K][A_CT], lp[K][A_CT];
  int act[K];
  bool valid[K];
  float kl = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int t = lane * K + k;
    valid[k] = t < Tm;
    rho[k] = 1.f; dsc[k] = 0.f; v[k] = 0.f; r[k] = 0.f; tlp[k] = 0.f; H[k] = 0.f; act[k] = 0;
#pragma unroll
    for (int j = 0; j < A_CT; ++j) { p[k][j] = 0.f; lp[k][j] = 0.f; }
    if (t < T) {
      const int64_t i = (int64_t)t * in_t + b * in_b;
      float blp[A_CT];
      log_softmax_row<A_CT>(tlog + i * A_CT, lp[k]);
      log_softmax_row<A_CT>(blog + i * A_CT, blp);
      float h = 0.f;
#pragma unroll
      for (int j = 0; j < A_CT; ++j) {
        p[k][j] = expf(lp[k][j]);
        h -= p[k][j] * lp[k][j];
        kl += p[k][j] * (lp[k][j] - blp[j]);
      }
      H[k] = h;
      if (valid[k]) {
        int a = (int)actions[i];
        if (a < 0 || a >= A_CT) { *err = 1; a = 0; }
        act[k] = a;
        float ta = lp[k][0], ba = blp[0];
#pragma unroll
        for (int j = 1; j < A_CT; ++j) { ta = (j == a) ? lp[k][j] : ta; ba = (j == a) ? blp[j] : ba; }
        tlp[k] = ta;
        dsc[k] = dones[i] ? 0.f : gamma;
        rho[k] = expf(ta - ba);
        v[k] = val[i];
        r[k] = rew[i];
      }
    }
  }
  vtrace_wave_core<K>(rho, dsc, v, r, valid, lane, Tm, bootstrap, clip_rho, clip_pg, vst, pgv);
  float pi = 0.f, vf = 0.f, ent = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int t = lane * K + k;
    if (t >= T || !live) continue;
    const int64_t i = (int64_t)t * in_t + b * in_b;
    if (valid[k]) {
      const int64_t o = (int64_t)t * in_t + b * out_b;
      pg[o] = pgv[k];
      vs[o] = vst[k];
      const float dv = v[k] - vst[k];
#pragma unroll
      for (int j = 0; j < A_CT; ++j)
        glog[i * A_CT + j] = -pgv[k] * ((j == act[k] ? 1.f : 0.f) - p[k][j]) - ent_coeff * (p[k][j] * (lp[k][j] + H[k]));
      gval[i] = vf_coeff * dv;
      pi -= tlp[k] * pgv[k];
      vf += 0.5f * dv * dv;
      ent += H[k];
    } else {  // the bootstrap row: no transition, no gradient
#pragma unroll
      for (int j = 0; j < A_CT; ++j) glog[i * A_CT + j] = 0.f;
      gval[i] = 0.f;
    }
  }
  // wave reduction, then the 4 waves of the workgroup through LDS: one f64 atomic per term and
  // WORKGROUP (with one per wave, 4 x 1024 atomics on four addresses made this kernel 55 us)
  pi = wave_sum(pi); vf = wave_sum(vf); ent = wave_sum(ent); kl = wave_sum(live ? kl : 0.f);
  __shared__ float red[4][4];
  if (lane == 0) {
    const int w = threadIdx.x >> 6;
    red[w][0] = pi; red[w][1] = vf; red[w][2] = ent; red[w][3] = kl;
  }
  __syncthreads();
  if (threadIdx.x < 4)
    atomicAdd(sums + threadIdx.x, (double)red[0][threadIdx.x] + (double)red[1][threadIdx.x] +
                                      (double)red[2][threadIdx.x] + (double)red[3][threadIdx.x]);
}

// V-trace from log-probs (the reference function boundary), time-major, wave-per-sequence:
// the small-B path of parlhip_vtrace_f32 (reference shape T'=49, B=1024 is 1.4 MB).
template <int K>
__global__ __launch_bounds__(256) void vtrace_wave_kernel(
    const float* __restrict__ blp, const float* __restrict__ tlp,
    const float* __restrict__ disc, const float* __restrict__ rew,
    const float* __restrict__ val, const float* __restrict__ boot,
    float* __restrict__ vs, float* __restrict__ pg, int T, int B, float clip_rho,
    float clip_pg) {
  const int lane = threadIdx.x & 63;
  const int64_t b = ((int64_t)xcd_chunk_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x) >> 6;
  if (b >= B) return;
  const float bootstrap = boot[b];
  float rho[K], dsc[K], v[K], r[K], vst[K], pgv[K];
  bool valid[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int t = lane * K + k;
    valid[k] = t < T;
    rho[k] = 1.f; dsc[k] = 0.f; v[k] = 0.f; r[k] = 0.f;
    if (valid[k]) {
      const int64_t i = (int64_t)t * B + b;
      rho[k] = expf(tlp[i] - blp[i]);
      dsc[k] = disc[i];
      v[k] = val[i];
      r[k] = rew[i];
    }
  }
  vtrace_wave_core<K>(rho, dsc, v, r, valid, lane, T, bootstrap, clip_rho, clip_pg, vst, pgv);
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int t = lane * K + k;
    if (!valid[k]) continue;
    const int64_t o = (int64_t)t * B + b;
    pg[o] = pgv[k];
    vs[o] = vst[k];
  }
}

// ----------------------------------------------------------------------------------------
// GAE / n-step return, lane-per-sequence, time-major.
// ----------------------------------------------------------------------------------------
template <int VEC, int U, bool DONE_F32, int CONV>
__global__ __launch_bounds__(256) void gae_tm_kernel(
    const float* __restrict__ rew, const float* __restrict__ val,
    const void* __restrict__ dones_v, const float* __restrict__ next_value,
    const void* __restrict__ last_done_v, float* __restrict__ adv,
    float* __restrict__ ret, int T, int B, float gamma, float gl) {
  const int64_t b0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
  if (b0 >= B) return;
  const float* dones_f = (const float*)dones_v;
  const uint8_t* dones_u = (const uint8_t*)dones_v;
  float carry[VEC], v_next[VEC], nnt_next[VEC];
  {
    Vec<VEC> nv = Vec<VEC>::load(next_value + b0);
#pragma unroll
    for (int j = 0; j < VEC; ++j) { carry[j] = 0.f; v_next[j] = nv.v[j]; nnt_next[j] = 1.f; }
    if (CONV == PARLHIP_GAE_DONE_STARTS_STEP) {
      if (DONE_F32) {
        Vec<VEC> ld = Vec<VEC>::load((const float*)last_done_v + b0);
#pragma unroll
        for (int j = 0; j < VEC; ++j) nnt_next[j] = 1.0f - ld.v[j];
      } else {
        U8Vec<VEC> ld = U8Vec<VEC>::load((const uint8_t*)last_done_v + b0);
#pragma unroll
        for (int j = 0; j < VEC; ++j) nnt_next[j] = 1.0f - (float)ld.v[j];
      }
    }
  }
  auto step = [&](const Vec<VEC>& xr, const Vec<VEC>& xv, const float* dn, int64_t i) {
    Vec<VEC> oa, orr;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float r = xr.v[j], v = xv.v[j];
      if (CONV == PARLHIP_GAE_DONE_ENDS_STEP) {
        const bool done = dn[j] != 0.f;
        const float nv = done ? 0.f : v_next[j];
        const float td = r + gamma * nv - v;
        carry[j] = done ? td : td + gl * carry[j];
      } else {
        // storage.py:57-60, op order preserved (float32 numpy)
        const float nnt = nnt_next[j];
        const float delta = r + gamma * v_next[j] * nnt - v;
        carry[j] = delta + gl * nnt * carry[j];
        nnt_next[j] = 1.0f - dn[j];   // 1 - dones[t] feeds step t-1
      }
      oa.v[j] = carry[j];
      orr.v[j] = carry[j] + v;
      v_next[j] = v;
    }
    if (adv) oa.store(adv + i);
    if (ret) orr.store(ret + i);
  };
  int t = T - 1;
  for (; t - (U - 1) >= 0; t -= U) {
    Vec<VEC> l_r[U], l_v[U];
    float l_d[U][VEC];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = (int64_t)(t - u) * B + b0;
      l_r[u] = Vec<VEC>::load(rew + i);
      l_v[u] = Vec<VEC>::load(val + i);
      if (DONE_F32) {
        Vec<VEC> d = Vec<VEC>::load(dones_f + i);
#pragma unroll
        for (int j = 0; j < VEC; ++j) l_d[u][j] = d.v[j];
      } else {
        U8Vec<VEC> d = U8Vec<VEC>::load(dones_u + i);
#pragma unroll
        for (int j = 0; j < VEC; ++j) l_d[u][j] = (float)d.v[j];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) step(l_r[u], l_v[u], l_d[u], (int64_t)(t - u) * B + b0);
  }
  for (; t >= 0; --t) {
    const int64_t i = (int64_t)t * B + b0;
    Vec<VEC> xr = Vec<VEC>::load(rew + i), xv = Vec<VEC>::load(val + i);
    float dn[VEC];
    if (DONE_F32) {
      Vec<VEC> d = Vec<VEC>::load(dones_f + i);
#pragma unroll
      for (int j = 0; j < VEC; ++j) dn[j] = d.v[j];
    } else {
      U8Vec<VEC> d = U8Vec<VEC>::load(dones_u + i);
#pragma unroll
      for (int j = 0; j < VEC; ++j) dn[j] = (float)d.v[j];
    }
    step(xr, xv, dn, i);
  }
}

// ----------------------------------------------------------------------------------------
// GAE for long T / small B (PPO config: T=2048, B=4096 -> only 64 lane-per-sequence waves, each
// a chain of 2048 dependent steps: latency-bound at ~0.8 TB/s).  The recurrence
//   adv_t = d_t + a_t * adv_{t+1}
// is affine, so T is cut into C chunks of kGaeChunk steps that run in parallel — in ONE pass over
// HBM: a workgroup (256 sequences x one chunk) loads its chunk once into registers (d_t, a_t, V_t),
// publishes the chunk's aggregate (A = prod a_t, D = chunk-local suffix value at its first step),
// folds the aggregates of all LATER chunks (x = D_k + A_k * x, k = C-1 .. c+1: the value of adv at
// the first step of chunk c+1) and finishes from registers.  20 B / element (f32 dones) instead of
// the 32 B of an aggregate pass + a final pass; the folded aggregates (8 B per sequence and chunk)
// come from L2.  Results differ from the single-pass kernel only by fp32 re-association of the
// carry (<= 1e-6 relative), within the 1e-5 contract.
// Forward progress: workgroups are numbered so that a chunk only ever waits for LOWER workgroup
// ids (later chunks are launched first), the usual decoupled look-back argument: the lowest
// unfinished id never waits for an unscheduled workgroup.
// ----------------------------------------------------------------------------------------
constexpr int kGaeChunk = 32;

template <bool DONE_F32, int CONV>
__global__ __launch_bounds__(256) void gae_lookback_kernel(
    const float* __restrict__ rew, const float* __restrict__ val,
    const void* __restrict__ dones_v, const float* __restrict__ next_value,
    const void* __restrict__ last_done_v, float* __restrict__ adv, float* __restrict__ ret,
    int T, int B, float gamma, float gl, int C, int cols, unsigned long long* __restrict__ ws) {
  constexpr int L = kGaeChunk;
  const int col = blockIdx.x % cols;
  const int c = C - 1 - (int)(blockIdx.x / cols);   // id 0 .. cols-1 = the last chunk in time
  const int64_t b = (int64_t)col * blockDim.x + threadIdx.x;
  const bool live = b < B;
  const int t0 = c * L;
  const int t1 = (t0 + L < T) ? t0 + L : T;
  const float* dones_f = (const float*)dones_v;
  const uint8_t* dones_u = (const uint8_t*)dones_v;
  auto done_at = [&](int64_t i) -> float { return DONE_F32 ? dones_f[i] : (float)dones_u[i]; };
  float dd[L], aa[L], vv[L];
  float carry = 0.f, Aprod = 1.f;
  if (live) {
    float v_next, nnt_next = 1.f;
    if (t1 == T) {
      v_next = next_value[b];
      if (CONV == PARLHIP_GAE_DONE_STARTS_STEP)
        nnt_next = 1.0f - (DONE_F32 ? ((const float*)last_done_v)[b] : (float)((const uint8_t*)last_done_v)[b]);
    } else {
      v_next = val[(int64_t)t1 * B + b];
      if (CONV == PARLHIP_GAE_DONE_STARTS_STEP) nnt_next = 1.0f - done_at((int64_t)t1 * B + b);
    }
    float lr[L], ld[L];
#pragma unroll
    for (int u = 0; u < L; ++u) {  // all loads of the chunk in flight before the dependent chain
      const int t = t1 - 1 - u;
      const int64_t i = (int64_t)(t < t0 ? t0 : t) * B + b;
      lr[u] = rew[i];
      vv[u] = val[i];
      ld[u] = done_at(i);
    }
#pragma unroll
    for (int u = 0; u < L; ++u) {
      const int t = t1 - 1 - u;
      float a = 1.f, d = 0.f;  // identity for steps before t0 (ragged last chunk in launch order)
      if (t >= t0) {
        const float r = lr[u], v = vv[u];
        if (CONV == PARLHIP_GAE_DONE_ENDS_STEP) {
          const bool done = ld[u] != 0.f;
          const float nv = done ? 0.f : v_next;
          d = r + gamma * nv - v;
          a = done ? 0.f : gl;
        } else {
          const float nnt = nnt_next;
          d = r + gamma * v_next * nnt - v;
          a = gl * nnt;
          nnt_next = 1.0f - ld[u];
        }
        carry = d + a * carry;
        Aprod = a * Aprod;
        v_next = v;
      }
      dd[u] = d;
      aa[u] = a;
    }
    // publish (A, D) as ONE 64-bit agent-scope atomic store: the workspace is pre-filled with the
    // all-ones pattern, so "A half != 0xffffffff" doubles as the ready flag — no separate flag, no
    // device-scope fence (on the 8-XCD part a release fence writes back the whole L2 of the XCD:
    // measured 290 us for this kernel with flags + __threadfence()).
    // a NaN whose bits are all ones (reachable from garbage float dones) would read as "never
    // published" and hang the dependent workgroups: publish the canonical quiet NaN instead
    unsigned abits = __float_as_uint(Aprod);
    abits = abits == 0xffffffffu ? 0x7fc00000u : abits;
    const unsigned long long packed = (unsigned long long)abits |
                                      ((unsigned long long)__float_as_uint(carry) << 32);
    __hip_atomic_store(ws + (int64_t)c * B + b, packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (!live) return;
  // fold the aggregates of the later chunks (x = adv at the first step of chunk c+1): batches of
  // independent 64-bit atomic loads; a batch with an unpublished entry is simply read again
  float x = 0.f;
  constexpr int KB = 8;
  for (int k = C - 1; k > c; k -= KB) {
    unsigned long long w[KB];
    for (;;) {
      bool ok = true;
#pragma unroll
      for (int j = 0; j < KB; ++j) {
        const int kk = (k - j > c) ? k - j : c + 1;   // clamped duplicates are skipped below
        w[j] = __hip_atomic_load(ws + (int64_t)kk * B + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok &= (unsigned)(w[j] & 0xffffffffull) != 0xffffffffu;
      }
      if (__ballot(!ok) == 0ull) break;   // wave-uniform retry keeps the wave converged
      __builtin_amdgcn_s_sleep(8);
    }
#pragma unroll
    for (int j = 0; j < KB; ++j)
      if (k - j > c) x = __uint_as_float((unsigned)(w[j] >> 32)) + __uint_as_float((unsigned)(w[j] & 0xffffffffull)) * x;
  }
  carry = x;
#pragma unroll
  for (int u = 0; u < L; ++u) {
    const int t = t1 - 1 - u;
    if (t < t0) continue;
    const int64_t i = (int64_t)t * B + b;
    // same expression shapes as gae_tm_kernel: done ? td : td + gl*carry  /  delta + gl*nnt*carry
    if (CONV == PARLHIP_GAE_DONE_ENDS_STEP) carry = (aa[u] == 0.f) ? dd[u] : dd[u] + gl * carry;
    else carry = dd[u] + aa[u] * carry;
    if (adv) adv[i] = carry;
    if (ret) ret[i] = carry + vv[u];
  }
}

template <int VEC, int U>
__global__ __launch_bounds__(256) void discount_cumsum_kernel(
    const float* __restrict__ x, const uint8_t* __restrict__ dones,
    float* __restrict__ out, int T, int B, float gamma) {
  const int64_t b0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
  if (b0 >= B) return;
  float carry[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) carry[j] = 0.f;
  int t = T - 1;
  for (; t >= 0; t -= U) {
    Vec<VEC> lx[U];
    U8Vec<VEC> ld[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (t - u < 0) continue;
      const int64_t i = (int64_t)(t - u) * B + b0;
      lx[u] = Vec<VEC>::load(x + i);
      if (dones) ld[u] = U8Vec<VEC>::load(dones + i);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (t - u < 0) continue;
      const int64_t i = (int64_t)(t - u) * B + b0;
      Vec<VEC> o;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const bool done = dones && ld[u].v[j];
        carry[j] = done ? lx[u].v[j] : lx[u].v[j] + gamma * carry[j];
        o.v[j] = carry[j];
      }
      o.store(out + i);
    }
  }
}

// per-thread error flag used by the logits kernels (bad action index)
__device__ int g_action_err;

// address of the device-side data-error word (bad action / minibatch index), resolved once per
// process; shared with ppo_kernels.hip
int* device_error_flag() {
  static int* p = nullptr;
  if (!p) {
    int* q = nullptr;
    if (check(hipGetSymbolAddress((void**)&q, HIP_SYMBOL(g_action_err))) != PARLHIP_OK)
      return nullptr;
    p = q;
  }
  return p;
}

}  // namespace parlhip

using namespace parlhip;

static int* action_err_ptr() { return device_error_flag(); }

PARLHIP_EXPORT int parlhip_consume_device_errors(parlhip_stream_t stream) {
  int* p = action_err_ptr();
  if (!p) return PARLHIP_ELAUNCH;
  int h = 0;
  hipStream_t s = (hipStream_t)stream;
  if (check(hipMemcpyAsync(&h, p, sizeof(int), hipMemcpyDeviceToHost, s))) return PARLHIP_ELAUNCH;
  if (check(hipStreamSynchronize(s))) return PARLHIP_ELAUNCH;
  if (h) {
    if (check(hipMemsetAsync(p, 0, sizeof(int), s))) return PARLHIP_ELAUNCH;
    if (check(hipStreamSynchronize(s))) return PARLHIP_ELAUNCH;
  }
  return h;
}

// Largest B for which time-major input takes the wave-per-sequence kernels (B waves): above it
// lane-per-sequence (B/64 waves of fully coalesced loads) has enough waves to hide latency.
static constexpr int kWaveSeqMaxB = 8192;

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Choose VEC: float4 lanes need B % 4 == 0, 16-B aligned bases, and enough sequences that a
// quarter as many lanes still fill the chip (>= 2 waves per SIMD over 256 CUs).
static inline bool use_vec4(int B, std::initializer_list<const void*> ptrs) {
  if (B % 4 != 0) return false;
  if ((int64_t)B / 4 < (int64_t)kNumCU * 4 * kWave * 2) return false;
  for (const void* p : ptrs)
    if (p && !aligned16(p)) return false;
  return true;
}

PARLHIP_EXPORT int parlhip_vtrace_f32(const float* blp, const float* tlp,
                                  const float* discounts, const float* rewards,
                                  const float* values, const float* bootstrap, float* vs,
                                  float* pg, int T, int B, float clip_rho, float clip_pg,
                                  parlhip_stream_t stream) {
  if (T < 0 || B < 0) return PARLHIP_EINVAL;
  if (T == 0 || B == 0) return PARLHIP_OK;
  if (!blp || !tlp || !discounts || !rewards || !values || !bootstrap || !vs || !pg)
    return PARLHIP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (B <= kWaveSeqMaxB && T <= 64 * 32) {
    // too few sequences for lane-per-sequence to fill 256 CUs: one wavefront per sequence
    const int K = ceil_div(T, 64);
    const int grid = ceil_div((int64_t)B * 64, 256);
#define LAUNCH_W(KK)                                                                      \
  vtrace_wave_kernel<KK><<<grid, 256, 0, s>>>(blp, tlp, discounts, rewards, values,      \
                                              bootstrap, vs, pg, T, B, clip_rho, clip_pg)
    if (K <= 1) LAUNCH_W(1);
    else if (K <= 2) LAUNCH_W(2);
    else if (K <= 4) LAUNCH_W(4);
    else if (K <= 8) LAUNCH_W(8);
    else if (K <= 16) LAUNCH_W(16);
    else LAUNCH_W(32);
#undef LAUNCH_W
    return check_launch();
  }
  // Launch shape measured on MI355X at T'=127 (tools/vt_variants.py, profiles/r01d_vtrace_variants.log):
  // outputs are written with nontemporal stores (they are not re-read by this kernel: +3 %);
  // float2 lanes x 4 time steps in flight win once B/2 lanes still fill the chip (5.67 vs 5.20 TB/s
  // for float4 x 4 at B = 1 M), scalar lanes x 8 steps win below (5.56 TB/s at B = 262,144).
  bool al8 = B % 2 == 0;
  for (const void* p : {(const void*)blp, (const void*)tlp, (const void*)discounts, (const void*)rewards,
                        (const void*)values, (const void*)bootstrap, (const void*)vs, (const void*)pg})
    al8 = al8 && (reinterpret_cast<uintptr_t>(p) % 8 == 0);
  if (al8 && (int64_t)B / 2 >= (int64_t)kNumCU * 4 * kWave * 4) {
    vtrace_tm_kernel<2, 4, true><<<ceil_div(B / 2, 256), 256, 0, s>>>(
        blp, tlp, discounts, rewards, values, bootstrap, vs, pg, T, B, clip_rho, clip_pg);
  } else {
    const int block = B >= 256 * 64 ? 256 : 64;
    vtrace_tm_kernel<1, 8, true><<<ceil_div(B, block), block, 0, s>>>(
        blp, tlp, discounts, rewards, values, bootstrap, vs, pg, T, B, clip_rho, clip_pg);
  }
  return check_launch();
}

template <int A_CT>
static int launch_vtrace_logits(const float* blog, const float* tlog, const int64_t* actions,
                                const float* rew, const uint8_t* dones, const float* val,
                                float* vs, float* pg, float* tlp_out, float* blp_out, int T,
                                int B, int A, int time_major, float gamma, float clip_rho,
                                float clip_pg, hipStream_t s, int* err) {
  const int Tm = T - 1;
  const int K = ceil_div(Tm, 64);
  if (time_major && (B > kWaveSeqMaxB || K > 32)) {
    const int block = B >= 256 * 64 ? 256 : 64;
    vtrace_logits_tm_kernel<A_CT><<<ceil_div(B, block), block, 0, s>>>(
        blog, tlog, actions, rew, dones, val, vs, pg, tlp_out, blp_out, T, B, A, gamma,
        clip_rho, clip_pg, err);
    return check_launch();
  }
  const int block = 256;  // 4 sequences per workgroup
  const int grid = ceil_div((int64_t)B * 64, block);
#define LAUNCH_EM(KK)                                                                    \
  do {                                                                                   \
    if (time_major)                                                                      \
      vtrace_logits_wave_kernel<A_CT, KK, true><<<grid, block, 0, s>>>(                  \
          blog, tlog, actions, rew, dones, val, vs, pg, tlp_out, blp_out, T, B, A,       \
          gamma, clip_rho, clip_pg, err);                                                \
    else                                                                                 \
      vtrace_logits_wave_kernel<A_CT, KK, false><<<grid, block, 0, s>>>(                 \
          blog, tlog, actions, rew, dones, val, vs, pg, tlp_out, blp_out, T, B, A,       \
          gamma, clip_rho, clip_pg, err);                                                \
  } while (0)
  if (K <= 1) LAUNCH_EM(1);
  else if (K <= 2) LAUNCH_EM(2);
  else if (K <= 4) LAUNCH_EM(4);
  else if (K <= 8) LAUNCH_EM(8);
  else if (K <= 16) LAUNCH_EM(16);
  else if (K <= 32) LAUNCH_EM(32);
  else return PARLHIP_ENOSUP;  // T > 2049 env-major: transpose to time-major instead
#undef LAUNCH_EM
  return check_launch();
}

PARLHIP_EXPORT int parlhip_vtrace_from_logits_f32(
    const float* blog, const float* tlog, const int64_t* actions, const float* rew,
    const uint8_t* dones, const float* val, float* vs, float* pg, float* tlp_out,
    float* blp_out, int T, int B, int A, int time_major, float gamma, float clip_rho,
    float clip_pg, parlhip_stream_t stream) {
  if (T < 1 || B < 0 || A < 1) return PARLHIP_EINVAL;
  if (T == 1 || B == 0) return PARLHIP_OK;
  if (!blog || !tlog || !actions || !rew || !dones || !val || !vs || !pg)
    return PARLHIP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  int* err = action_err_ptr();
  if (!err) return PARLHIP_ELAUNCH;
#define ARGS blog, tlog, actions, rew, dones, val, vs, pg, tlp_out, blp_out, T, B, A, \
             time_major, gamma, clip_rho, clip_pg, s, err
  switch (A) {
    case 2: return launch_vtrace_logits<2>(ARGS);
    case 3: return launch_vtrace_logits<3>(ARGS);
    case 4: return launch_vtrace_logits<4>(ARGS);
    case 6: return launch_vtrace_logits<6>(ARGS);
    case 9: return launch_vtrace_logits<9>(ARGS);
    case 18: return launch_vtrace_logits<18>(ARGS);
    default: return launch_vtrace_logits<0>(ARGS);
  }
#undef ARGS
}

template <bool DONE_F32, int CONV>
static int launch_gae(const float* rew, const float* val, const void* dones,
                      const float* next_value, const void* last_done, float* adv, float* ret,
                      int T, int B, float gamma, float gl, hipStream_t s) {
  bool v4 = use_vec4(B, {rew, val, next_value, adv, ret});
  if (v4) {
    const uintptr_t dal = DONE_F32 ? 15 : 3;
    if ((reinterpret_cast<uintptr_t>(dones) & dal) ||
        (last_done && (reinterpret_cast<uintptr_t>(last_done) & dal)))
      v4 = false;
  }
  if (v4) {
    gae_tm_kernel<4, 4, DONE_F32, CONV><<<ceil_div(B / 4, 256), 256, 0, s>>>(
        rew, val, dones, next_value, last_done, adv, ret, T, B, gamma, gl);
  } else {
    const int block = B >= 256 * 64 ? 256 : 64;
    gae_tm_kernel<1, 8, DONE_F32, CONV><<<ceil_div(B, block), block, 0, s>>>(
        rew, val, dones, next_value, last_done, adv, ret, T, B, gamma, gl);
  }
  return check_launch();
}

PARLHIP_EXPORT int parlhip_gae_f32(const float* rew, const float* val, const void* dones,
                               const float* next_value, const void* last_done, float* adv,
                               float* ret, int T, int B, float gamma, float lam,
                               int done_convention, int dones_are_f32,
                               parlhip_stream_t stream) {
  if (T < 0 || B < 0) return PARLHIP_EINVAL;
  if (T == 0 || B == 0) return PARLHIP_OK;
  if (!rew || !val || !dones || !next_value || (!adv && !ret)) return PARLHIP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (done_convention == PARLHIP_GAE_DONE_ENDS_STEP) {
    const float gl = gamma * lam;
    return dones_are_f32
               ? launch_gae<true, PARLHIP_GAE_DONE_ENDS_STEP>(rew, val, dones, next_value,
                                                              nullptr, adv, ret, T, B, gamma, gl, s)
               : launch_gae<false, PARLHIP_GAE_DONE_ENDS_STEP>(rew, val, dones, next_value,
                                                               nullptr, adv, ret, T, B, gamma, gl, s);
  }
  if (done_convention == PARLHIP_GAE_DONE_STARTS_STEP) {
    if (!last_done) return PARLHIP_EINVAL;
    // Python: gamma * gae_lambda is a double product, rounded to f32 at the array op
    const float gl = (float)((double)gamma * (double)lam);
    return dones_are_f32
               ? launch_gae<true, PARLHIP_GAE_DONE_STARTS_STEP>(rew, val, dones, next_value,
                                                                last_done, adv, ret, T, B, gamma, gl, s)
               : launch_gae<false, PARLHIP_GAE_DONE_STARTS_STEP>(rew, val, dones, next_value,
                                                                 last_done, adv, ret, T, B, gamma, gl, s);
  }
  return PARLHIP_EINVAL;
}

// Chunk plan for the long-T / small-B path: 0 chunks = stay on the single-pass kernel.
static inline void gae_chunk_plan(int T, int B, int* C_out, int* cols_out) {
  *C_out = 0; *cols_out = 0;
  const int waves = ceil_div(B, 64);
  if (T < 256 || waves >= 1024) return;   // enough lane-per-sequence waves to fill the chip already
  *C_out = ceil_div(T, kGaeChunk);
  *cols_out = ceil_div(B, 256);
}

PARLHIP_EXPORT size_t parlhip_gae_workspace_bytes(int T, int B) {
  int C, cols;
  if (T <= 0 || B <= 0) return 0;
  gae_chunk_plan(T, B, &C, &cols);
  return (size_t)C * (size_t)B * sizeof(unsigned long long);   // one (A, D) pair per chunk and sequence
}

template <bool DONE_F32, int CONV>
static int launch_gae_chunked(const float* rew, const float* val, const void* dones,
                              const float* next_value, const void* last_done, float* adv, float* ret,
                              int T, int B, float gamma, float gl, int C, int cols, float* ws,
                              hipStream_t s) {
  // all-ones = "not published yet" (see gae_lookback_kernel)
  int rc = check(hipMemsetAsync(ws, 0xff, (size_t)C * B * sizeof(unsigned long long), s));
  if (rc) return rc;
  gae_lookback_kernel<DONE_F32, CONV><<<C * cols, 256, 0, s>>>(rew, val, dones, next_value, last_done, adv, ret, T,
                                                               B, gamma, gl, C, cols, (unsigned long long*)ws);
  return check_launch();
}

PARLHIP_EXPORT int parlhip_gae_ws_f32(const float* rew, const float* val, const void* dones,
                                  const float* next_value, const void* last_done, float* adv,
                                  float* ret, int T, int B, float gamma, float lam,
                                  int done_convention, int dones_are_f32, void* workspace,
                                  size_t workspace_bytes, parlhip_stream_t stream) {
  int C = 0, L = 0;
  if (T > 0 && B > 0) gae_chunk_plan(T, B, &C, &L);
  if (C == 0)
    return parlhip_gae_f32(rew, val, dones, next_value, last_done, adv, ret, T, B, gamma, lam,
                           done_convention, dones_are_f32, stream);
  if (!rew || !val || !dones || !next_value || (!adv && !ret)) return PARLHIP_EINVAL;
  if (!workspace || workspace_bytes < parlhip_gae_workspace_bytes(T, B)) return PARLHIP_ENOMEM;
  hipStream_t s = (hipStream_t)stream;
  float* ws = (float*)workspace;
  if (done_convention == PARLHIP_GAE_DONE_ENDS_STEP) {
    const float gl = gamma * lam;
    return dones_are_f32 ? launch_gae_chunked<true, PARLHIP_GAE_DONE_ENDS_STEP>(
                               rew, val, dones, next_value, nullptr, adv, ret, T, B, gamma, gl, C, L, ws, s)
                         : launch_gae_chunked<false, PARLHIP_GAE_DONE_ENDS_STEP>(
                               rew, val, dones, next_value, nullptr, adv, ret, T, B, gamma, gl, C, L, ws, s);
  }
  if (done_convention == PARLHIP_GAE_DONE_STARTS_STEP) {
    if (!last_done) return PARLHIP_EINVAL;
    const float gl = (float)((double)gamma * (double)lam);
    return dones_are_f32 ? launch_gae_chunked<true, PARLHIP_GAE_DONE_STARTS_STEP>(
                               rew, val, dones, next_value, last_done, adv, ret, T, B, gamma, gl, C, L, ws, s)
                         : launch_gae_chunked<false, PARLHIP_GAE_DONE_STARTS_STEP>(
                               rew, val, dones, next_value, last_done, adv, ret, T, B, gamma, gl, C, L, ws, s);
  }
  return PARLHIP_EINVAL;
}

PARLHIP_EXPORT int parlhip_discount_cumsum_f32(const float* x, const uint8_t* dones, float* out,
                                           int T, int B, float gamma,
                                           parlhip_stream_t stream) {
  if (T < 0 || B < 0) return PARLHIP_EINVAL;
  if (T == 0 || B == 0) return PARLHIP_OK;
  if (!x || !out) return PARLHIP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  bool v4 = use_vec4(B, {x, out}) && !(reinterpret_cast<uintptr_t>(dones) & 3);
  if (v4) {
    discount_cumsum_kernel<4, 4><<<ceil_div(B / 4, 256), 256, 0, s>>>(x, dones, out, T, B, gamma);
  } else {
    const int block = B >= 256 * 64 ? 256 : 64;
    discount_cumsum_kernel<1, 8><<<ceil_div(B, block), block, 0, s>>>(x, dones, out, T, B, gamma);
  }
  return check_launch();
}

template <int A_CT>
static int launch_impala_loss(const float* blog, const float* tlog, const int64_t* actions, const float* rew,
                              const uint8_t* dones, const float* val, float* vs, float* pg, float* glog,
                              float* gval, double* sums, int T, int B, int time_major, float gamma,
                              float clip_rho, float clip_pg, float vf_coeff, float ent_coeff, hipStream_t s,
                              int* err) {
  const int K = ceil_div(T, 64);   // T rows (the bootstrap row takes part in the KL sum)
  const int grid = ceil_div((int64_t)B * 64, 256);
#define LAUNCH_LOSS(KK)                                                                                  \
  do {                                                                                                   \
    if (time_major)                                                                                      \
      impala_loss_wave_kernel<A_CT, KK, true><<<grid, 256, 0, s>>>(blog, tlog, actions, rew, dones, val, \
          vs, pg, glog, gval, sums, T, B, gamma, clip_rho, clip_pg, vf_coeff, ent_coeff, err);           \
    else                                                                                                 \
      impala_loss_wave_kernel<A_CT, KK, false><<<grid, 256, 0, s>>>(blog, tlog, actions, rew, dones, val,\
          vs, pg, glog, gval, sums, T, B, gamma, clip_rho, clip_pg, vf_coeff, ent_coeff, err);           \
  } while (0)
  if (K <= 1) LAUNCH_LOSS(1);
  else if (K <= 2) LAUNCH_LOSS(2);
  else if (K <= 4) LAUNCH_LOSS(4);
  else return PARLHIP_ENOSUP;   // T > 256: use parlhip_vtrace_from_logits_f32 + the framework's loss
#undef LAUNCH_LOSS
  return check_launch();
}

PARLHIP_EXPORT int parlhip_impala_loss_f32(const float* blog, const float* tlog, const int64_t* actions,
                                           const float* rew, const uint8_t* dones, const float* val, float* vs,
                                           float* pg, float* grad_logits, float* grad_values, double* sums, int T,
                                           int B, int A, int time_major, float gamma, float clip_rho,
                                           float clip_pg, float vf_coeff, float ent_coeff,
                                           parlhip_stream_t stream) {
  if (T < 2 || B < 0 || A < 1) return PARLHIP_EINVAL;
  if (B == 0) return PARLHIP_OK;
  if (!blog || !tlog || !actions || !rew || !dones || !val || !vs || !pg || !grad_logits || !grad_values || !sums)
    return PARLHIP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  int* err = action_err_ptr();
  if (!err) return PARLHIP_ELAUNCH;
#define LARGS blog, tlog, actions, rew, dones, val, vs, pg, grad_logits, grad_values, sums, T, B, time_major, \
              gamma, clip_rho, clip_pg, vf_coeff, ent_coeff, s, err
  switch (A) {
    case 2: return launch_impala_loss<2>(LARGS);
    case 3: return launch_impala_loss<3>(LARGS);
    case 4: return launch_impala_loss<4>(LARGS);
    case 6: return launch_impala_loss<6>(LARGS);
    case 9: return launch_impala_loss<9>(LARGS);
    case 18: return launch_impala_loss<18>(LARGS);
    default: return PARLHIP_ENOSUP;   // other action counts: the unfused path
  }
#undef LARGS
}

// ----------------------------------------------------------------------------------------
// IMPALA learner: both heads + the whole V-trace loss + the heads' backward in ONE kernel.
//
// What impala_loss_wave_kernel fuses (log-softmax, gather, entropy, KL, V-trace, loss sums, gradient
// w.r.t. logits / values) moves 5 MB at the reference shape (T=50, B=1024, A=6): launch-bound by
// construction.  The tensors next to it are not small: the trunk output h [T*B, 256] f32 (52 MB) is
// read by the two head GEMMs (policy_fc, value_fc: atari_model.py:44-57) and by their weight-gradient
// GEMMs, and d total / d h (52 MB) is written by their input-gradient GEMMs.  Here the rows of a sequence are
// read ONCE into registers, the A+1 head outputs of every row are formed, the loss math runs lane-per-step
// exactly as impala_loss_wave_kernel does, and d total / d h and the heads' weight / bias gradients are
// produced from the rows still held.  Algorithmic bytes per (t, b) row: 1024 (h) + 1024 (dh) + 4A (behaviour
// logits) + 8 + 4 + 1 in, 8 out for T-1 rows  =>  107 MB per launch at the reference shape instead of 5 MB.
// Weight-gradient partials: one per workgroup, fixed-order sum (heads_partial_sum_kernel): deterministic.
// Time-major, T <= 64, 256 hidden units.  (Rounds 2-4 ran this with TWO waves per sequence, 128 columns each,
// the rows summed by a 64-lane butterfly and the output gradients broadcast with v_readlane: ~3,800 VALU
// instructions per wave, two waves per SIMD, 32-33 us; the layout below replaced it in round 5.)
// ----------------------------------------------------------------------------------------
template <int A_CT>
__device__ __forceinline__ void log_softmax_regs(const float (&x)[A_CT], float (&lp)[A_CT]) {
  float m = x[0];
#pragma unroll
  for (int j = 1; j < A_CT; ++j) m = fmaxf(m, x[j]);
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < A_CT; ++j) sum += expf(x[j] - m);
  const float lse = logf(sum);
#pragma unroll
  for (int j = 0; j < A_CT; ++j) lp[j] = (x[j] - m) - lse;
}

__device__ __forceinline__ float lane_bcast(float x, int src_lane) {  // src_lane: compile-time after unrolling
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), src_lane));
}

constexpr int kHeadsHidden = 256;

// gfx950 lane swaps (the clang builtins of this toolchain return the same value twice — checked on the
// device — hence the instruction itself; a VALU write needs two wait states before the swap reads it)
__device__ __forceinline__ void lane_swap32(float& a, float& c) {  // a[32..63] <-> c[0..31]
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(c));
}
__device__ __forceinline__ void lane_swap16(float& a, float& c) {  // odd 16-lane rows of a <-> even rows of c
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(c));
}
typedef float f2v __attribute__((ext_vector_type(2)));
template <int DPP_CTRL>
__device__ __forceinline__ float dpp_partner_add(float keep, float send) {
  return keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), DPP_CTRL, 0xf,
                                                                      0xf, true));
}
template <int DPP_CTRL>
__device__ __forceinline__ float stage_combine(float x, float y, bool hi) {
  return dpp_partner_add<DPP_CTRL>(hi ? y : x, hi ? x : y);
}
// ----------------------------------------------------------------------------------------
// FOUR wavefronts per sequence, rows across the lane groups (round 5).
//
// A wave owns 64 columns of its sequence and lane (n = lane & 15, kk = lane >> 4) holds columns
// 64 w + 4 n .. + 3 (one float4) of the rows t = 4 G + kk, G = 0 .. NG-1: one dwordx4 load per row group (four
// rows x 256 contiguous bytes), NG = 13 registers-of-four for T <= 52.
//  * head outputs: the lane's 4-column dot product per (G, j), then a butterfly over the 16 lanes of a DPP
//    row only (the four lane groups hold DIFFERENT rows: nothing to reduce across them) — stages 8 and 4 as two
//    bank-masked v_add_f32_dpp each (row_mirror / row_half_mirror: the lanes that keep a value are whole banks),
//    stages 2 and 1 as select + quad_perm add — over 16 leaves (2 row groups x 8 output slots) per result
//    register, depth first, in row-group order (= load order: the waits for the loads are progressive).
//    The four column quarters meet in LDS, ONE wave per sequence runs the loss math lane-per-step (its inputs
//    fetched and the behaviour policy's log-softmax taken while h is in flight; the V-trace recurrence as a DPP
//    scan, no LDS-crossbar shuffles) and hands d total / d (logits, value) back through LDS.
//  * backward: d h = g W on the MATRIX pipe (v_mfma_f32_16x16x4_f32 per 16-row tile and column component: the
//    four components' result registers are float4s of four rows, stored 4 rows x 256 B per instruction), d W =
//    g^T h on the vector ALUs (every lane group reads ITS row's 8 gradients with two ds_read_b128 — a broadcast
//    inside the group, no v_readlane — and does 14 v_pk_fma_f32 per row group); the waves of the workgroup's two
//    sequences run the two passes in opposite order so that both pipes of a SIMD are busy.  The four lane groups'
//    weight gradients are folded with v_permlane32_swap / v_permlane16_swap (two registers per swap + add); the
//    reductions over the steps (bias gradients, loss sums) are taken by two other waves from LDS with DPP adds.
// Two sequences (eight waves) per workgroup, one partial per workgroup.  <= 128 VGPRs: four waves per SIMD
// alone, two beside the emulator (the two-wave kernel: 164 VGPRs, one).
// Measured at (50, 1024, 6) (rocprofv3, kernel only): 32.1 us -> 25.7 us (0.40 -> 0.52 of the HBM peak); beside
// the emulator 66 -> 47 us.  Where the time goes (diagnostic builds, PARLHIP_HQ_ABL): h in + d h out alone
// 16.8 us (6.4 TB/s), + head outputs 1.8, + loss 3.3, + backward 3.8 — the phases of a workgroup are separated by
// barriers and every workgroup of the grid is resident at once, so they add instead of overlapping.
// ----------------------------------------------------------------------------------------
typedef float f4v __attribute__((ext_vector_type(4)));

// wave-wide sum without the LDS crossbar: four DPP adds inside each row of 16 lanes (quad_perm, row_half_mirror,
// row_mirror: every lane ends with its row's total), then the four row totals through the scalar unit
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
  return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
}

template <int CTRL>
__device__ __forceinline__ float dpp_or(float fallback, float x) {  // x of the DPP source lane, `fallback` where there is none
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fallback), __builtin_bit_cast(int, x),
                                                               CTRL, 0xf, 0xf, false));
}
// value of lane + 1 (wave_shl:1; lane 63 gets `last`)
__device__ __forceinline__ float next_lane(float x, float last) { return dpp_or<0x130>(last, x); }

// vtrace_wave_core<1> (one step per lane) without ds_bpermute: the suffix composition of the affine maps
// acc_t = d_t + a_t acc_{t+1} is a Hillis-Steele scan inside each row of 16 lanes with row_shl:1,2,4,8 (a lane
// without a source keeps the identity map: `old` operand, bound_ctrl off), then the three row totals (lane 0 of
// each row) are composed through v_readlane.  Same arithmetic, a different association of the products than the
// 64-lane shuffle scan: equal within float rounding (tests: 1e-5 against the CPU oracle).
__device__ __forceinline__ void vtrace_wave_core_dpp(float rho, float dsc, float v, float r, bool valid, int lane, int Tm,
                                                     float bootstrap, float clip_rho, float clip_pg, float& vst,
                                                     float& pgv) {
  float v_next = next_lane(v, 0.f);
  if (lane + 1 >= Tm) v_next = bootstrap;
  const float crho = clip_max(rho, clip_rho);
  const float c = fminf(rho, 1.0f);
  const float d_ = valid ? crho * (r + dsc * v_next - v) : 0.f;
  const float a_ = valid ? dsc * c : 1.f;
  float SA = a_, SD = d_;
#define HQ_SCAN_STEP(CTRL)                      \
  do {                                          \
    const float oa = dpp_or<CTRL>(1.f, SA);     \
    const float od = dpp_or<CTRL>(0.f, SD);     \
    SD = SD + SA * od;                          \
    SA = SA * oa;                               \
  } while (0)
  HQ_SCAN_STEP(0x101);  // row_shl:1
  HQ_SCAN_STEP(0x102);  // row_shl:2
  HQ_SCAN_STEP(0x104);  // row_shl:4
  HQ_SCAN_STEP(0x108);  // row_shl:8
#undef HQ_SCAN_STEP
  // carry entering row q = total(row q+1) o total(row q+2) o ... (D part only: acc beyond the wave is 0)
  const float A1 = lane_bcast(SA, 16), D1 = lane_bcast(SD, 16), A2 = lane_bcast(SA, 32), D2 = lane_bcast(SD, 32);
  const float D3 = lane_bcast(SD, 48);
  const float c2 = D3, c1 = D2 + A2 * c2, c0 = D1 + A1 * c1;
  const int row = lane >> 4;
  const float cin = row == 0 ? c0 : (row == 1 ? c1 : (row == 2 ? c2 : 0.f));
  const float suffix = SD + SA * cin;   // acc_t of this lane's step
  vst = suffix + v;
  float vs_next = next_lane(vst, 0.f);
  if (lane + 1 >= Tm) vs_next = bootstrap;
  pgv = clip_max(rho, clip_pg) * (r + dsc * vs_next - v);
}

template <int A_CT, int NG, int I>
__device__ __forceinline__ float hq_leaf(const f4v (&hr)[NG], const f4v (&wj)[A_CT + 1]) {
  constexpr int G = I >> 3, j = I & 7;
  if constexpr (G < NG && j <= A_CT) {
    f2v p = f2v{hr[G].x, hr[G].y} * f2v{wj[j].x, wj[j].y};
    p = __builtin_elementwise_fma(f2v{hr[G].z, hr[G].w}, f2v{wj[j].z, wj[j].w}, p);
    return p.x + p.y;
  } else {
    return 0.f;
  }
}
// Stages 8 and 4 of the butterfly without selects (BM): the lanes that keep x are whole DPP banks (4 lanes) —
// banks 0,1 | 2,3 at stage 8, banks 0,2 | 1,3 at stage 4 — so "x + partner's x where I keep x, y + partner's y
// where I keep y" is two v_add_f32_dpp with complementary bank masks writing one register (2 instead of 3 ops).
template <int OFF>
__device__ __forceinline__ float stage_combine_banks(float x, float y) {
  float r;
  if constexpr (OFF == 8)
    asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %2, %2 row_mirror row_mask:0xf bank_mask:0xc" : "=&v"(r) : "v"(x), "v"(y));
  else
    asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %0, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xa" : "=&v"(r) : "v"(x), "v"(y));
  return r;
}
// sum over the 16 lanes of a DPP row of 16 leaves: lane n ends with the row sum of leaf I + n
template <int A_CT, int NG, int OFF, int I>
__device__ __forceinline__ float hq_tree(const f4v (&hr)[NG], const f4v (&wj)[A_CT + 1], int lane) {
  if constexpr (OFF == 16) {
    return hq_leaf<A_CT, NG, I>(hr, wj);
  } else {
    const float x = hq_tree<A_CT, NG, OFF * 2, I>(hr, wj, lane);
    const float y = hq_tree<A_CT, NG, OFF * 2, I + OFF>(hr, wj, lane);
    if constexpr (OFF >= 4) {
      return stage_combine_banks<OFF>(x, y);
    } else {
      constexpr int ctrl = OFF == 8 ? 0x140 : (OFF == 4 ? 0x141 : (OFF == 2 ? 0x4e : 0xb1));
      return stage_combine<ctrl>(x, y, (lane & OFF) != 0);
    }
  }
}

// diagnostic builds (tools/build_obj_variant.sh hqN scan_kernels.hip -DPARLHIP_HQ_ABL=N): 1 copy h -> d h only,
// 2 no loss math, 3 no forward trees, 4 no backward arithmetic (d h = h), 5 no weight-gradient pass
#ifndef PARLHIP_HQ_ABL
#define PARLHIP_HQ_ABL 0
#endif
// WIDE: a name tag only (same code) — launches over >= 256 sequences (the workload shape, 107 MB) and the 20-sequence
// launches of a 1000-row update (2 MB, launch-bound) appear as two rows in a rocprofv3 kernel summary instead of one
// average over both.
template <int A_CT, int NG, bool WIDE>  // 4 NG >= T rows
__global__ __launch_bounds__(512, 4) void impala_heads_loss_q_kernel(
    const float* __restrict__ h, const float* __restrict__ wpi, const float* __restrict__ bpi,
    const float* __restrict__ wv, const float* __restrict__ bv, const float* __restrict__ blog,
    const int64_t* __restrict__ actions, const float* __restrict__ rew, const uint8_t* __restrict__ dones,
    float* __restrict__ vs, float* __restrict__ pg, float* __restrict__ dh, float* __restrict__ wpart,
    double* __restrict__ sums, int T, int B, float gamma, float clip_rho, float clip_pg, float vf_coeff,
    float ent_coeff, int* __restrict__ err) {
  constexpr int NO = A_CT + 1, H = kHeadsHidden, NR = 4 * NG, NQ = (NG + 1) / 2, NRP = 8 * NQ;  // NRP: rows incl. the padding group
  static_assert(NO <= 8, "eight output slots per row");
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int w = wid & 3, sq = wid >> 2;
  const int n = lane & 15, kk = lane >> 4;
  const int blk = xcd_chunk_block(blockIdx.x, gridDim.x);
  const int64_t b_raw = (int64_t)blk * 2 + sq;
  const bool live = b_raw < B;
  const int64_t b = live ? b_raw : (int64_t)B - 1;
  const int Tm = T - 1;
  __shared__ __attribute__((aligned(16))) float xl[2][4][NRP * 8];  // [sequence][column quarter][row][output slot]
  __shared__ __attribute__((aligned(16))) float gl[2][NR * 8];      // [sequence][row][output slot]: d total / d output
  __shared__ __attribute__((aligned(16))) float redw[2][NO][H];
  __shared__ float redb[2][NO + 4];
  __shared__ __attribute__((aligned(16))) float sl[2][64][4];      // [sequence][step]: pi, vf, entropy, kl terms

  // (A start delay for half of the workgroups — so that one cohort's loads stream in under the other's arithmetic —
  // was measured with this layout too: 26.4 us at 0, 27.2 / 28.5 / 30.2 us at 3.4 / 5.1 / 6.8 us of delay, for grid
  // halves and for alternating workgroups alike.  Removed.)
  // ---- 1. loads: row group G = rows 4G + kk, this wave's 64 columns.  Rows past T-1 re-read row T-1: their head
  // outputs are never used, their output gradients are zero (so they add nothing to the weight gradients) and
  // their d h rows are not stored — no zero fill, no divergent branch.
  const size_t BH = (size_t)B * H;
  const float* hb = h + b * H + 64 * w + 4 * n;
  f4v hr[NG];
#pragma unroll
  for (int G = 0; G < NG; ++G) hr[G] = *(const f4v*)(hb + (size_t)min(4 * G + kk, Tm) * BH);
  if constexpr (PARLHIP_HQ_ABL == 1) {
    float* dq = dh + b * H + 64 * w + 4 * n;
#pragma unroll
    for (int G = 0; G < NG; ++G)
      if (live && 4 * G + kk < T) *(f4v*)(dq + (size_t)(4 * G + kk) * BH) = hr[G];
    return;
  }
  f4v wj[NO];
#pragma unroll
  for (int j = 0; j < NO; ++j) wj[j] = *(const f4v*)((j < A_CT ? wpi + (size_t)j * H : wv) + 64 * w + 4 * n);
  // The loss wave of the sequence fetches ITS inputs now (behaviour logits, action, reward, done of step t = lane)
  // and takes the behaviour policy's log-softmax while the rows of h are still in flight: after the forward
  // these loads would be an exposed HBM round trip with the whole workgroup waiting at the barrier behind it.
  const bool loss_wave = w == ((int)b_raw & 3);   // spread over the SIMDs
  const int t = lane;
  const bool in_t = t < T, valid1 = t < Tm;
  const int64_t i = (int64_t)(in_t ? t : 0) * B + b;
  float blp[A_CT], bias[NO], r_in = 0.f, dsc_in = 0.f;
  int act = 0;
#pragma unroll
  for (int j = 0; j < A_CT; ++j) blp[j] = 0.f;
#pragma unroll
  for (int j = 0; j < NO; ++j) bias[j] = 0.f;
  if (loss_wave) {
#pragma unroll
    for (int j = 0; j < NO; ++j) {  // scalar loads: pinned here, not behind the barrier where they are used
      bias[j] = j < A_CT ? bpi[j] : bv[0];
      asm volatile("" : "+s"(bias[j]));
    }
    log_softmax_row<A_CT>(blog + i * A_CT, blp);
    if (valid1) {
      int a = (int)actions[i];
      if (a < 0 || a >= A_CT) { *err = 1; a = 0; }
      act = a;
      dsc_in = dones[i] ? 0.f : gamma;
      r_in = rew[i];
    }
  }
  // ---- 2. head outputs of this column quarter: result register q, lane (n, kk) = row 4 (2q + (n >> 3)) + kk, slot n & 7
  {
    float* xw = &xl[sq][w][32 * (n >> 3) + 8 * kk + (n & 7)];
#define HQ_TREE(Q) if constexpr (Q < NQ && PARLHIP_HQ_ABL != 3) { xw[64 * Q] = hq_tree<A_CT, NG, 1, 16 * Q>(hr, wj, lane); }
    HQ_TREE(0) HQ_TREE(1) HQ_TREE(2) HQ_TREE(3) HQ_TREE(4) HQ_TREE(5) HQ_TREE(6) HQ_TREE(7)
#undef HQ_TREE
  }
  __syncthreads();
  // ---- 3. the loss on ONE wave per sequence, lane per step (impala_loss_wave_kernel, K = 1).  Only what the
  // other waves wait for — d total / d (logits, value) — comes before the barrier; the outputs and the sums after it.
  float g[NO];
#pragma unroll
  for (int j = 0; j < NO; ++j) g[j] = 0.f;
  float pi = 0.f, vf = 0.f, ent = 0.f, kl = 0.f, vs_t = 0.f, pg_t = 0.f;
  if (loss_wave && PARLHIP_HQ_ABL != 2) {
    const int tr = t < NR ? t : 0;
    float o8[8];
    {
      const f4v* x0 = (const f4v*)&xl[sq][0][tr * 8];
      const f4v* x1 = (const f4v*)&xl[sq][1][tr * 8];
      const f4v* x2 = (const f4v*)&xl[sq][2][tr * 8];
      const f4v* x3 = (const f4v*)&xl[sq][3][tr * 8];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f4v sm = (x0[q] + x1[q]) + (x2[q] + x3[q]);
        o8[4 * q] = sm.x; o8[4 * q + 1] = sm.y; o8[4 * q + 2] = sm.z; o8[4 * q + 3] = sm.w;
      }
    }
    float outv[NO];
#pragma unroll
    for (int j = 0; j < NO; ++j) outv[j] = o8[j] + bias[j];
    const float v_own = outv[A_CT];
    const float bootstrap = lane_bcast(v_own, Tm);
    float lp[A_CT], p[A_CT];
    {
      float tl[A_CT];
#pragma unroll
      for (int j = 0; j < A_CT; ++j) tl[j] = outv[j];
      log_softmax_regs<A_CT>(tl, lp);
    }
    float Hh = 0.f;
#pragma unroll
    for (int j = 0; j < A_CT; ++j) {
      p[j] = expf(lp[j]);
      Hh -= p[j] * lp[j];
      kl += p[j] * (lp[j] - blp[j]);
    }
    if (!in_t || !live) kl = 0.f;
    float rho = 1.f, v_t = 0.f, tlp = 0.f, vst, pgv;
    if (valid1) {
      float ta = lp[0], ba = blp[0];
#pragma unroll
      for (int j = 1; j < A_CT; ++j) { ta = (j == act) ? lp[j] : ta; ba = (j == act) ? blp[j] : ba; }
      tlp = ta;
      rho = expf(ta - ba);
      v_t = v_own;
    }
    vtrace_wave_core_dpp(rho, dsc_in, v_t, r_in, valid1, lane, Tm, bootstrap, clip_rho, clip_pg, vst, pgv);
    if (valid1 && live) {
      const float dv = v_t - vst;
#pragma unroll
      for (int j = 0; j < A_CT; ++j)
        g[j] = -pgv * ((j == act ? 1.f : 0.f) - p[j]) - ent_coeff * (p[j] * (lp[j] + Hh));
      g[A_CT] = vf_coeff * dv;
      pg_t = pgv;
      vs_t = vst;
      pi = -tlp * pgv;
      vf = 0.5f * dv * dv;
      ent = Hh;
    }
    if (t < NR) {
      f4v* go = (f4v*)&gl[sq][t * 8];
      go[0] = f4v{g[0], NO > 1 ? g[NO > 1 ? 1 : 0] : 0.f, NO > 2 ? g[NO > 2 ? 2 : 0] : 0.f, NO > 3 ? g[NO > 3 ? 3 : 0] : 0.f};
      go[1] = f4v{NO > 4 ? g[NO > 4 ? 4 : 0] : 0.f, NO > 5 ? g[NO > 5 ? 5 : 0] : 0.f, NO > 6 ? g[NO > 6 ? 6 : 0] : 0.f,
                  NO > 7 ? g[NO > 7 ? 7 : 0] : 0.f};
    }
    *(f4v*)&sl[sq][t][0] = f4v{pi, vf, ent, kl};
  }
  __syncthreads();
  // after the barrier every wave goes into its backward; the loss wave only stores its two outputs, and the
  // reductions over the steps (bias gradients, loss sums) are taken by the sequence's next two waves from LDS
  if (loss_wave && valid1 && live) {
    const int64_t o = (int64_t)t * B + b;
    pg[o] = pg_t;
    vs[o] = vs_t;
  }
  if (w == (((int)b_raw + 1) & 3)) {        // d total / d bias_j = sum_t g[t][j]
    const f4v* gp = (const f4v*)&gl[sq][(t < NR ? t : 0) * 8];
    f4v a0 = gp[0], a1 = gp[1];
    if (t >= NR) { a0 = f4v{0.f, 0.f, 0.f, 0.f}; a1 = a0; }
    const float gs[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
    for (int j = 0; j < NO; ++j) {
      const float sm = wave_sum_dpp(gs[j]);
      if (lane == 0) redb[sq][j] = sm;
    }
  } else if (w == (((int)b_raw + 2) & 3)) {  // pi_loss, vf_loss, entropy, kl
    const f4v x = *(const f4v*)&sl[sq][t][0];
    const float s0 = wave_sum_dpp(x.x), s1 = wave_sum_dpp(x.y), s2 = wave_sum_dpp(x.z), s3 = wave_sum_dpp(x.w);
    if (lane == 0) { redb[sq][NO] = s0; redb[sq][NO + 1] = s1; redb[sq][NO + 2] = s2; redb[sq][NO + 3] = s3; }
  }
  // ---- 4. backward.  d h = g W on the matrix pipe, d W = g^T h on the vector ALUs; the waves of the workgroup's
  // first sequence run d W first, those of the second d h first, so that every SIMD (two waves of each kind) has
  // both pipes busy (as VALU code alone the two passes were 4.7 us of this kernel: 364 v_pk_fma per wave, four waves
  // per SIMD).
  //   d h: v_mfma_f32_16x16x4_f32 per tile of 16 rows and per column component c (column 64 w + 4 n + c):
  //        A[i = row][k = j] = g[16 tau + i][4 s + k] (one ds_read_b32 per k-step s = 0, 1), B[k = j][n] = W_j[column],
  //        D[row 4 (lane >> 4) + r][n]: the four components' register r are one float4 of row 16 tau + 4 (lane >> 4) + r.
  //   d W: per row group its row's 8 gradients from LDS (two ds_read_b128, a broadcast inside the lane group) and
  //        14 v_pk_fma_f32 on the lane's four columns.
  f2v acc[NO][2];
  auto pass_dw = [&]() {
#pragma unroll
    for (int j = 0; j < NO; ++j) acc[j][0] = acc[j][1] = f2v{0.f, 0.f};
#pragma unroll
    for (int G = 0; G < ((PARLHIP_HQ_ABL == 4 || PARLHIP_HQ_ABL == 5) ? 0 : NG); ++G) {
      const f4v* gp = (const f4v*)&gl[sq][(4 * G + kk) * 8];
      const f4v g0 = gp[0], g1 = gp[1];
      const float gr[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const f2v h0 = {hr[G].x, hr[G].y}, h1 = {hr[G].z, hr[G].w};
#pragma unroll
      for (int j = 0; j < NO; ++j) {
        const f2v gg = {gr[j], gr[j]};
        acc[j][0] = __builtin_elementwise_fma(gg, h0, acc[j][0]);
        acc[j][1] = __builtin_elementwise_fma(gg, h1, acc[j][1]);
      }
    }
  };
  auto pass_dh = [&]() {
    const float* wq = wpi;  // opaque copies: the weights are RE-loaded here (L1 / L2 hits) instead of living across the loss
    const float* vq = wv;
    asm volatile("" : "+s"(wq), "+s"(vq));
    f4v wB[2];   // k-step s: W_j, j = min(4 s + kk, A) (the value head is output A; slots past it carry zero gradients)
#pragma unroll
    for (int sk = 0; sk < 2; ++sk) {
      const int j = min(4 * sk + kk, A_CT);
      wB[sk] = *(const f4v*)((j < A_CT ? wq + (size_t)j * H : vq) + 64 * w + 4 * n);
    }
    float* dhb = dh + b * H + 64 * w + 4 * n + (size_t)(4 * kk) * BH;   // row 4 kk of a tile
    constexpr int NT = (NG + 3) / 4;
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) {
      f4v D[4];
      if constexpr (PARLHIP_HQ_ABL != 4) {
        const int ar = min(16 * tau + n, NR - 1) * 8 + kk;
        const float a0 = gl[sq][ar], a1 = gl[sq][ar + 4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          D[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, wB[0][c], f4v{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          D[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, wB[1][c], D[c], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) D[c] = hr[0];
      }
      float* dt = dhb + (size_t)(16 * tau) * BH;
      if (live && 16 * tau + 16 <= T) {   // wave-uniform: a whole tile, four unconditional stores
#pragma unroll
        for (int r = 0; r < 4; ++r) *(f4v*)(dt + (size_t)r * BH) = f4v{D[0][r], D[1][r], D[2][r], D[3][r]};
      } else if (live) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (16 * tau + 4 * kk + r < T) *(f4v*)(dt + (size_t)r * BH) = f4v{D[0][r], D[1][r], D[2][r], D[3][r]};
      }
    }
  };
  // (the fence keeps the second pass's LDS reads behind the first pass: hoisted above it they spilled 90 registers)
  // (gl's address is handed to the asm: a __shared__ array nobody takes the address of is invisible to a "memory" clobber)
#define HQ_PASS_FENCE() asm volatile("" : : "v"(&gl[0][0]) : "memory")
  if (sq == 0) {
    pass_dw();
    // the accumulators are pinned here: the arithmetic of the d W pass otherwise SINKS below the d h pass to its
    // first use (the fold), with the 91 gradient registers it reads kept alive across the MFMAs (spills)
#pragma unroll
    for (int j = 0; j < NO; ++j) asm volatile("" : "+v"(acc[j][0]), "+v"(acc[j][1]));
    HQ_PASS_FENCE();
    pass_dh();
  } else {
    pass_dh();
    HQ_PASS_FENCE();
    pass_dw();
  }
#undef HQ_PASS_FENCE
  // ---- 5. fold the four lane groups (rows mod 4): register r = 7 c + j holds column component c of output j;
  // swap32 + add folds registers r | r + 2 NO (lanes < 32 keep r), swap16 + add r | r + NO: lane group kk ends
  // with component c = kk of every output: d W[j][64 w + 4 n + kk]
  {
    float v[4 * NO];
#pragma unroll
    for (int j = 0; j < NO; ++j) {
      v[j] = acc[j][0].x; v[NO + j] = acc[j][0].y; v[2 * NO + j] = acc[j][1].x; v[3 * NO + j] = acc[j][1].y;
    }
    float u[2 * NO];
#pragma unroll
    for (int i = 0; i < 2 * NO; ++i) {
      lane_swap32(v[i], v[i + 2 * NO]);
      u[i] = v[i] + v[i + 2 * NO];
    }
#pragma unroll
    for (int j = 0; j < NO; ++j) {
      lane_swap16(u[j], u[j + NO]);
      redw[sq][j][64 * w + 4 * n + kk] = u[j] + u[j + NO];
    }
  }
  __syncthreads();
  float* wp = wpart + (size_t)blk * (NO * H + NO);
  for (int k = threadIdx.x; k < NO * (H / 4); k += 512) {
    const f4v a0 = ((const f4v*)&redw[0][0][0])[k], a1 = ((const f4v*)&redw[1][0][0])[k];
    ((f4v*)wp)[k] = a0 + a1;
  }
  if (threadIdx.x < NO) wp[NO * H + threadIdx.x] = redb[0][threadIdx.x] + redb[1][threadIdx.x];
  if (threadIdx.x < 4) atomicAdd(sums + threadIdx.x, (double)redb[0][NO + threadIdx.x] + (double)redb[1][NO + threadIdx.x]);
}

// out[k] = sum over workgroup partials, fixed order.  This launch is a chain of dependent HBM round trips (the
// partials were written by other XCDs), so what counts is how many rounds a thread needs: 16 outputs x 16 slices
// per block (partial_sum16: eight rounds of four loads at 512 partials) took 4.3-5 us; with 8 outputs x 32
// slices and up to 16 loads of a thread in flight at once it is one round for <= 512 partials.
__global__ __launch_bounds__(256) void heads_partial_sum_kernel(const float* __restrict__ wpart, int nblk, int n,
                                                                float* __restrict__ out) {
  __shared__ float red[256];
  const int kk = threadIdx.x & 7, qs = threadIdx.x >> 3;   // output within the block, slice 0..31
  const int k = blockIdx.x * 8 + kk;
  float acc = 0.f;
  if (k < n) {
    for (int q0 = qs; q0 < nblk; q0 += 32 * 16) {
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int q = q0 + 32 * i;
        v[i] = q < nblk ? wpart[(size_t)q * n + k] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) acc += v[i];
    }
  }
  red[qs * 8 + kk] = acc;
  __syncthreads();
  if (qs == 0 && k < n) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) t += red[i * 8 + kk];
    out[k] = t;
  }
}

PARLHIP_EXPORT size_t parlhip_impala_heads_loss_workspace_bytes(int B, int A) {
  if (B <= 0 || A <= 0) return 0;
  return (size_t)ceil_div(B, 2) * ((size_t)(A + 1) * kHeadsHidden + (A + 1)) * sizeof(float);
}

PARLHIP_EXPORT int parlhip_impala_heads_loss_f32(const float* hidden, const float* w_policy, const float* b_policy,
                                                 const float* w_value, const float* b_value,
                                                 const float* behaviour_logits, const int64_t* actions,
                                                 const float* rewards, const uint8_t* dones, float* vs, float* pg,
                                                 float* grad_hidden, float* grad_heads, double* sums,
                                                 void* workspace, int T, int B, int hidden_units, int A, float gamma,
                                                 float clip_rho, float clip_pg, float vf_coeff, float ent_coeff,
                                                 parlhip_stream_t stream) {
  if (T < 2 || B < 0 || A < 1) return PARLHIP_EINVAL;
  if (T > 64 || hidden_units != kHeadsHidden) return PARLHIP_ENOSUP;
  if (B == 0) return PARLHIP_OK;
  if (!hidden || !w_policy || !b_policy || !w_value || !b_value || !behaviour_logits || !actions || !rewards ||
      !dones || !vs || !pg || !grad_hidden || !grad_heads || !sums || !workspace)
    return PARLHIP_EINVAL;
  if ((reinterpret_cast<uintptr_t>(hidden) | reinterpret_cast<uintptr_t>(grad_hidden) |
       reinterpret_cast<uintptr_t>(w_policy) | reinterpret_cast<uintptr_t>(w_value) |
       reinterpret_cast<uintptr_t>(workspace)) & 15)
    return PARLHIP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  int* err = action_err_ptr();
  if (!err) return PARLHIP_ELAUNCH;
  const int nblk = ceil_div(B, 2);  // two sequences (four half-sequence waves) per workgroup
  float* wpart = (float*)workspace;
#define HQT(AA, NGG, WW)                                                                                       \
  impala_heads_loss_q_kernel<AA, NGG, WW><<<nblk, 512, 0, s>>>(hidden, w_policy, b_policy, w_value, b_value,  \
      behaviour_logits, actions, rewards, dones, vs, pg, grad_hidden, wpart, sums, T, B, gamma, clip_rho,     \
      clip_pg, vf_coeff, ent_coeff, err)
  if (B > (1 << 20)) return PARLHIP_ENOSUP;  // 32-bit lane offsets of the row-group layout
#define HL(AA) do { if (B >= 256) { if (T <= 52) HQT(AA, 13, true); else HQT(AA, 16, true); } else { if (T <= 52) HQT(AA, 13, false); else HQT(AA, 16, false); } } while (0)
  switch (A) {
    case 4: HL(4); break;
    case 6: HL(6); break;
    default: return PARLHIP_ENOSUP;  // other action counts: parlhip_impala_loss_f32 behind the framework's heads
  }
#undef HL
#undef HQT
  const int n = (A + 1) * kHeadsHidden + (A + 1);
  heads_partial_sum_kernel<<<ceil_div(n, 8), 256, 0, s>>>(wpart, nblk, n, grad_heads);
  return check_launch();
}
