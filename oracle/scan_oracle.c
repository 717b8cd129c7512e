/*
 * scan_oracle.c — CPU restatement (plain C) of the reference's scan / sampling arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker.  The product path (parl_amd/) never imports it.
 *
 * Every function cites the reference lines (relative to the PARL tree) it follows.
 * Pinning: tests/test_oracle_golden.py checks these functions against
 *   - the known-answer recipe of parl/algorithms/paddle/impala/tests/vtrace_test_paddle.py:33-144,
 *   - outputs of the reference's own calc_gae / RolloutStorage.compute_returns (run in the
 *     build container, committed under tests/golden/ by tests/golden/make_golden.py),
 *   - numpy's own np.random.choice for the sampler.
 *
 * Compile with -ffp-contract=off: the float32 paths must round exactly like numpy.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define IDX(t, b) ((size_t)(t) * (size_t)B + (size_t)(b))

/* parl/algorithms/paddle/impala/vtrace.py:36-139 (float32 throughout, like paddle).
 * NaN threshold == the reference's `None` (no clipping).                               */
int oracle_vtrace_f32(const float* blp, const float* tlp, const float* discounts,
                      const float* rewards, const float* values, const float* bootstrap,
                      float* vs, float* pg_adv, int T, int B, float clip_rho,
                      float clip_pg_rho) {
  if (T < 0 || B < 0) return -1;
  for (int b = 0; b < B; ++b) {
    float acc = 0.0f;                  /* vtrace.py:116 acc = zeros_like(bootstrap) */
    float vs_next = bootstrap[b];      /* vtrace.py:128-129 vs_t_plus_1[-1] = bootstrap */
    float v_next = bootstrap[b];       /* vtrace.py:110-111 values_t_plus_1[-1]        */
    for (int t = T - 1; t >= 0; --t) {
      size_t i = IDX(t, b);
      float log_rho = tlp[i] - blp[i];                    /* :99  */
      float rho = expf(log_rho);                          /* :101 */
      float crho = isnan(clip_rho) ? rho : fminf(rho, clip_rho);      /* :102-105 */
      float c = fminf(rho, 1.0f);                         /* :107 */
      float disc = discounts[i];
      float v = values[i];
      float r = rewards[i];
      float delta = crho * (r + disc * v_next - v);       /* :114 */
      acc = delta + disc * c * acc;                       /* :119 */
      float vs_t = acc + v;                               /* :125 */
      float cpg = isnan(clip_pg_rho) ? rho : fminf(rho, clip_pg_rho); /* :131-134 */
      pg_adv[i] = cpg * (r + disc * vs_next - v);         /* :136-137 */
      vs[i] = vs_t;
      vs_next = vs_t;
      v_next = v;
    }
  }
  return 0;
}

/* log-softmax gather: IMPALA._log_prob, impala.py:119-132 = sum(log_softmax(logits)*onehot).
 * float32, max-subtracted like paddle/torch log_softmax.                               */
static float log_prob_f32(const float* logits, int A, int64_t a) {
  float m = logits[0];
  for (int k = 1; k < A; ++k) m = fmaxf(m, logits[k]);
  float s = 0.0f;
  for (int k = 0; k < A; ++k) s += expf(logits[k] - m);
  return (logits[a] - m) - logf(s);
}

/* impala.py:119-132 (_log_prob ×2), :59 (discounts), :167-194 (split/drop-last/bootstrap)
 * + vtrace.py:36-139.  Layout rules are those of include/parl_hip.h.                    */
int oracle_vtrace_from_logits_f32(const float* blogits, const float* tlogits,
                                  const int64_t* actions, const float* rewards,
                                  const uint8_t* dones, const float* values, float* vs,
                                  float* pg_adv, float* tlp_out, float* blp_out, int T,
                                  int B, int A, int time_major, float gamma,
                                  float clip_rho, float clip_pg_rho) {
  if (T < 1 || B < 0 || A < 1) return -1;
  const int Tm = T - 1;
  for (int b = 0; b < B; ++b) {
    size_t last = time_major ? ((size_t)(T - 1) * B + b) : ((size_t)b * T + (T - 1));
    float bootstrap = values[last];    /* impala.py:191-194 */
    float acc = 0.0f, vs_next = bootstrap, v_next = bootstrap;
    for (int t = Tm - 1; t >= 0; --t) {
      size_t i = time_major ? ((size_t)t * B + b) : ((size_t)b * T + t);
      size_t o = time_major ? ((size_t)t * B + b) : ((size_t)b * Tm + t);
      int64_t a = actions[i];
      if (a < 0 || a >= A) return -1;
      float tlp = log_prob_f32(tlogits + i * A, A, a);
      float blp = log_prob_f32(blogits + i * A, A, a);
      float disc = dones[i] ? 0.0f : gamma;   /* impala.py:59 (~dones).astype(f32)*discount */
      float rho = expf(tlp - blp);
      float crho = isnan(clip_rho) ? rho : fminf(rho, clip_rho);
      float c = fminf(rho, 1.0f);
      float v = values[i], r = rewards[i];
      float delta = crho * (r + disc * v_next - v);
      acc = delta + disc * c * acc;
      float vs_t = acc + v;
      float cpg = isnan(clip_pg_rho) ? rho : fminf(rho, clip_pg_rho);
      pg_adv[o] = cpg * (r + disc * vs_next - v);
      vs[o] = vs_t;
      if (tlp_out) tlp_out[o] = tlp;
      if (blp_out) blp_out[o] = blp;
      vs_next = vs_t;
      v_next = v;
    }
  }
  return 0;
}

/* GAE.
 * done_convention 0: examples/A2C/actor.py:73-85 + parl/utils/rl_utils.py:34-51.  A segment
 *   closes at done or at the last step; next_value = 0 if done else V(next_obs)
 *   (actor.py:75-78); within a segment adv_t = td_t + gamma*lam*adv_{t+1} (rl_utils.py:31,49-50).
 *   The reference evaluates this in float64 (lfilter); accum_f64 selects that, otherwise the
 *   float32 arithmetic the HIP kernel uses.
 * done_convention 1: examples/PPO/storage.py:45-64, float32 numpy, op order preserved:
 *   delta = r + gamma*nextvalues*nextnonterminal - v
 *   adv = lastgaelam = delta + gamma*gae_lambda*nextnonterminal*lastgaelam
 *   (gamma*gae_lambda is a Python double product rounded to f32 when it meets the array). */
int oracle_gae_f32(const float* rewards, const float* values, const void* dones,
                   const float* next_value, const void* last_done, float* adv,
                   float* ret, int T, int B, float gamma, float lam, int done_convention,
                   int dones_are_f32, int accum_f64) {
  if (T < 0 || B < 0) return -1;
  const uint8_t* d8 = (const uint8_t*)dones;
  const float* df = (const float*)dones;
  const uint8_t* l8 = (const uint8_t*)last_done;
  const float* lf = (const float*)last_done;
  if (done_convention == 0) {
    for (int b = 0; b < B; ++b) {
      if (accum_f64) {
        double carry = 0.0, v_next = (double)next_value[b];
        const double g = (double)gamma, gl = (double)gamma * (double)lam;
        for (int t = T - 1; t >= 0; --t) {
          size_t i = IDX(t, b);
          int done = dones_are_f32 ? (df[i] != 0.0f) : (d8[i] != 0);
          double nv = done ? 0.0 : v_next;
          double td = (double)rewards[i] + g * nv - (double)values[i];
          carry = done ? td : td + gl * carry;
          if (adv) adv[i] = (float)carry;
          if (ret) ret[i] = (float)(carry + (double)values[i]);
          v_next = (double)values[i];
        }
      } else {
        float carry = 0.0f, v_next = next_value[b];
        const float gl = gamma * lam;
        for (int t = T - 1; t >= 0; --t) {
          size_t i = IDX(t, b);
          int done = dones_are_f32 ? (df[i] != 0.0f) : (d8[i] != 0);
          float nv = done ? 0.0f : v_next;
          float td = rewards[i] + gamma * nv - values[i];
          carry = done ? td : td + gl * carry;
          if (adv) adv[i] = carry;
          if (ret) ret[i] = carry + values[i];
          v_next = values[i];
        }
      }
    }
    return 0;
  }
  if (done_convention == 1) {
    const float gl = (float)((double)gamma * (double)lam); /* python float product -> f32 */
    for (int b = 0; b < B; ++b) {
      float lastgaelam = 0.0f;
      for (int t = T - 1; t >= 0; --t) {
        size_t i = IDX(t, b);
        float nnt, nextv;
        if (t == T - 1) {                      /* storage.py:51-53 */
          float dl = dones_are_f32 ? lf[b] : (float)l8[b];
          nnt = 1.0f - dl;
          nextv = next_value[b];
        } else {                               /* storage.py:54-56 */
          size_t j = IDX(t + 1, b);
          float dn = dones_are_f32 ? df[j] : (float)d8[j];
          nnt = 1.0f - dn;
          nextv = values[j];
        }
        float delta = rewards[i] + gamma * nextv * nnt - values[i];   /* :57-58 */
        lastgaelam = delta + gl * nnt * lastgaelam;                    /* :59-60 */
        if (adv) adv[i] = lastgaelam;
        if (ret) ret[i] = lastgaelam + values[i];                      /* :61 */
      }
    }
    return 0;
  }
  return -1;
}

/* parl/utils/rl_utils.py:21-31: lfilter([1],[1,-gamma], x[::-1])[::-1], batched; optional
 * dones reset the carry after a terminal step.                                          */
int oracle_discount_cumsum_f32(const float* x, const uint8_t* dones, float* out, int T,
                               int B, float gamma, int accum_f64) {
  if (T < 0 || B < 0) return -1;
  for (int b = 0; b < B; ++b) {
    if (accum_f64) {
      double carry = 0.0;
      for (int t = T - 1; t >= 0; --t) {
        size_t i = IDX(t, b);
        carry = (dones && dones[i]) ? (double)x[i] : (double)x[i] + (double)gamma * carry;
        out[i] = (float)carry;
      }
    } else {
      float carry = 0.0f;
      for (int t = T - 1; t >= 0; --t) {
        size_t i = IDX(t, b);
        carry = (dones && dones[i]) ? x[i] : x[i] + gamma * carry;
        out[i] = carry;
      }
    }
  }
  return 0;
}

/* parl/algorithms/torch/ppo.py:115-117 / paddle ppo.py:124-127:
 * (adv - adv.mean()) / (adv.std() + 1e-8), std unbiased (N-1).  Sums in float64.        */
int oracle_adv_normalize_f32(const float* adv, const int64_t* idx, float* out, int64_t n,
                             float eps, float* mean_std_out) {
  if (n < 0) return -1;
  double s = 0.0;
  for (int64_t i = 0; i < n; ++i) s += (double)adv[idx ? idx[i] : i];
  double mean = n > 0 ? s / (double)n : 0.0;
  double q = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    double d = (double)adv[idx ? idx[i] : i] - mean;
    q += d * d;
  }
  double var = n > 1 ? q / (double)(n - 1) : NAN; /* torch/paddle: std of 1 element = nan */
  float fmean = (float)mean, fstd = (float)sqrt(var);
  for (int64_t i = 0; i < n; ++i)
    out[i] = (adv[idx ? idx[i] : i] - fmean) / (fstd + eps);
  if (mean_std_out) { mean_std_out[0] = fmean; mean_std_out[1] = fstd; }
  return 0;
}

/* np.random.choice(A, 1, p=prob) — examples/IMPALA/atari_agent.py:38-40.  numpy's legacy
 * RandomState.choice: p -> float64, cdf = p.cumsum(); cdf /= cdf[-1];
 * idx = cdf.searchsorted(u, side='right').                                             */
int oracle_categorical_sample_f32(const float* probs, const double* uniforms,
                                  int64_t* actions, int B, int A) {
  if (B < 0 || A < 1 || A > 4096) return -1;
  double cdf[4096];
  for (int b = 0; b < B; ++b) {
    const float* p = probs + (size_t)b * A;
    double s = 0.0;
    for (int k = 0; k < A; ++k) { s += (double)p[k]; cdf[k] = s; }
    const double last = cdf[A - 1];
    const double u = uniforms[b];
    int64_t a = A;                       /* searchsorted(side='right'): first cdf[k] > u */
    for (int k = 0; k < A; ++k) {
      if (cdf[k] / last > u) { a = k; break; }
    }
    actions[b] = a;
  }
  return 0;
}

/* Philox4x32-10 (Salmon et al., SC'11), the counter-based generator the HIP sampler uses.
 * key = (seed lo, seed hi); counter = (offset lo, offset hi, row lo, row hi).           */
static inline void philox_round(uint32_t c[4], const uint32_t k[2]) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
  uint32_t n1 = (uint32_t)p1;
  uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
  uint32_t n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
void oracle_philox4x32_10(uint64_t seed, uint64_t offset, uint64_t row, uint32_t out[4]) {
  uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t c[4] = {(uint32_t)offset, (uint32_t)(offset >> 32), (uint32_t)row,
                   (uint32_t)(row >> 32)};
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k);
    k[0] += 0x9E3779B9u;
    k[1] += 0xBB67AE85u;
  }
  memcpy(out, c, sizeof(uint32_t) * 4);
}
/* 53-bit uniform in [0,1): same construction as numpy's random_sample
 * ((a>>5)*2^26 + (b>>6)) / 2^53 from two 32-bit words.                                  */
double oracle_philox_uniform53(uint64_t seed, uint64_t offset, uint64_t row) {
  uint32_t w[4];
  oracle_philox4x32_10(seed, offset, row, w);
  uint64_t a = w[0] >> 5, b = w[1] >> 6;
  return (double)(a * 67108864ull + b) / 9007199254740992.0;
}

/* softmax as IMPALA.sample does (impala.py:226 F.softmax) in float32, then the choice above
 * with philox uniforms — the CPU twin of parlhip_policy_sample_f32.                      */
int oracle_policy_sample_f32(const float* x, int is_logits, int64_t* actions,
                             float* probs_out, double* uniforms_out, int B, int A,
                             uint64_t seed, uint64_t offset, uint64_t row0) {
  if (B < 0 || A < 1 || A > 4096) return -1;
  float p[4096];
  for (int b = 0; b < B; ++b) {
    const float* xr = x + (size_t)b * A;
    if (is_logits) {
      float m = xr[0];
      for (int k = 1; k < A; ++k) m = fmaxf(m, xr[k]);
      float s = 0.0f;
      for (int k = 0; k < A; ++k) { p[k] = expf(xr[k] - m); s += p[k]; }
      for (int k = 0; k < A; ++k) p[k] = p[k] / s;
    } else {
      memcpy(p, xr, sizeof(float) * A);
    }
    double u = oracle_philox_uniform53(seed, offset, row0 + (uint64_t)b);
    oracle_categorical_sample_f32(p, &u, actions + b, 1, A);
    if (probs_out) memcpy(probs_out + (size_t)b * A, p, sizeof(float) * A);
    if (uniforms_out) uniforms_out[b] = u;
  }
  return 0;
}
