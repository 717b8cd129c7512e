"""CPU baselines for the scan kernels — TEST / MEASUREMENT INFRASTRUCTURE ONLY
(imported by bench.py's cpu_baseline leg and by tests; never by parl_amd).

The reference computes these on the host in numpy / scipy / a Python loop of framework ops.  The
GPU box has no /root/reference, so each baseline is a numpy PORT with the reference's own loop
structure (kind "port"), pinned in tests/test_oracle_golden.py on fixtures produced by the
reference functions themselves; where /root/reference exists the reference function itself is
loaded by path instead (kind "reference").

  calc_gae_segments   parl/utils/rl_utils.py:21-51 called per (env, segment) as
                      examples/A2C/actor.py:73-85 does (lists of floats in, lfilter)
  compute_returns     examples/PPO/storage.py:45-64 (loop over T, vectorised over E)
  vtrace_numpy        parl/algorithms/paddle/impala/vtrace.py:99-137 (per-t loop; the reference runs
                      it as ~3 framework ops per t on the learner device)
"""
import importlib.util
import os
import sys
import time

import numpy as np
import scipy.signal

REF = '/root/reference'


def _calc_gae_port(rewards, values, next_value, gamma, lam):
    tds = rewards + gamma * np.append(values[1:], next_value) - values            # rl_utils.py:49
    return scipy.signal.lfilter([1.0], [1.0, -gamma * lam], tds[::-1])[::-1]        # :31,50


def load_calc_gae():
    p = os.path.join(REF, 'parl/utils/rl_utils.py')
    if os.path.exists(p):
        spec = importlib.util.spec_from_file_location('ref_rl_utils', p)
        m = importlib.util.module_from_spec(spec)
        keep, sys.dont_write_bytecode = sys.dont_write_bytecode, True  # no __pycache__ in the reference tree
        try:
            spec.loader.exec_module(m)
        finally:
            sys.dont_write_bytecode = keep
        return m.calc_gae, 'reference'
    return _calc_gae_port, 'port'


def calc_gae_segments(rewards, values, dones, next_value, gamma, lam, fn=None):
    """A2C actor semantics (actor.py:73-85): per env, cut at done or at the end of the rollout,
    next_value = 0 after a terminal step; returns advantages [T,E] (float64)"""
    fn = fn or _calc_gae_port
    T, E = rewards.shape
    adv = np.zeros((T, E), np.float64)
    for e in range(E):
        start = 0
        for t in range(T):
            if dones[t, e] or t == T - 1:
                nv = 0.0 if dones[t, e] else float(next_value[e])
                r = [float(x) for x in rewards[start:t + 1, e]]   # the actor accumulates Python lists
                v = [float(x) for x in values[start:t + 1, e]]
                adv[start:t + 1, e] = fn(r, v, nv, gamma, lam)
                start = t + 1
    return adv


def compute_returns(rewards, values, dones, value, done, gamma=0.99, gae_lambda=0.95):
    """storage.py:45-64 on [T,E] float32 arrays; dones[t] = "obs t starts a new episode" """
    T = rewards.shape[0]
    advantages = np.zeros_like(rewards)
    lastgaelam = 0
    for t in reversed(range(T)):
        if t == T - 1:
            nextnonterminal = 1.0 - done
            nextvalues = value.reshape(1, -1)
        else:
            nextnonterminal = 1.0 - dones[t + 1]
            nextvalues = values[t + 1]
        delta = rewards[t] + gamma * nextvalues * nextnonterminal - values[t]
        advantages[t] = lastgaelam = delta + gamma * gae_lambda * nextnonterminal * lastgaelam
    return advantages, advantages + values


def vtrace_numpy(blp, tlp, discounts, rewards, values, bootstrap, clip_rho=1.0, clip_pg_rho=1.0):
    """vtrace.py:99-137 in float32 numpy with the reference's per-t loop"""
    rhos = np.exp(tlp - blp)                                                        # :99-101
    clipped_rhos = np.minimum(rhos, clip_rho) if clip_rho is not None else rhos     # :102-105
    cs = np.minimum(rhos, 1.0)                                                      # :107
    values_t_plus_1 = np.concatenate([values[1:], bootstrap[None]], 0)             # :110-111
    deltas = clipped_rhos * (rewards + discounts * values_t_plus_1 - values)        # :112-114
    acc = np.zeros_like(bootstrap)                                                  # :116
    out = []
    for t in reversed(range(values.shape[0])):                                      # :118-122
        acc = deltas[t] + discounts[t] * cs[t] * acc
        out.append(acc)
    vs = np.stack(out[::-1]) + values                                               # :122-125
    vs_t_plus_1 = np.concatenate([vs[1:], bootstrap[None]], 0)                      # :128-129
    pg_rhos = np.minimum(rhos, clip_pg_rho) if clip_pg_rho is not None else rhos    # :131-134
    return vs, pg_rhos * (rewards + discounts * vs_t_plus_1 - values)               # :135-137


def _median_time(fn, runs, warm=2):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(runs):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def time_scan_baselines(seed=0):
    """single-core timings at the BASELINE shapes; returns a dict of {name: {...}}"""
    rng = np.random.default_rng(seed)
    out = {}
    # C2: A2C GAE, T=20, E=256, per-(env, segment) calls
    T, E = 20, 256
    rew = rng.choice([-1.0, 0.0, 1.0], p=[.01, .98, .01], size=(T, E)).astype(np.float32)
    val = rng.standard_normal((T, E)).astype(np.float32)
    dn = rng.random((T, E)) < 1 / 800
    nv = rng.standard_normal(E).astype(np.float32)
    fn, kind = load_calc_gae()
    s = _median_time(lambda: calc_gae_segments(rew, val, dn, nv, 0.99, 1.0, fn), 10)
    out['calc_gae_per_segment_T20_E256'] = {'seconds': s, 'elements_per_s': T * E / s, 'kind': kind, 'cores': 1,
                                            'what': 'rl_utils.calc_gae called per (env, segment), actor.py:73-85'}
    # C5: PPO compute_returns, T=2048, E=4096
    T, E = 2048, 4096
    rew = np.clip(rng.standard_normal((T, E)), -10, 10).astype(np.float32)
    val = rng.standard_normal((T, E)).astype(np.float32)
    dn = (rng.random((T, E)) < 1e-3).astype(np.float32)
    s = _median_time(lambda: compute_returns(rew, val, dn, val[0], dn[0]), 3, warm=1)
    out['ppo_compute_returns_T2048_E4096'] = {'seconds': s, 'elements_per_s': T * E / s, 'GBps': T * E * 20 / s / 1e9,
                                              'kind': 'port', 'cores': 1, 'what': 'storage.py:45-64 loop over T'}
    # C3: V-trace T'=49, B=1024 and a larger batch
    for Tq, B, runs in ((49, 1024, 20), (127, 65536, 3)):
        x = [rng.standard_normal((Tq, B)).astype(np.float32) for _ in range(5)]
        x[2] = np.full((Tq, B), 0.99, np.float32)
        boot = rng.standard_normal(B).astype(np.float32)
        s = _median_time(lambda: vtrace_numpy(x[0], x[1], x[2], x[3], x[4], boot), runs, warm=1)
        out['vtrace_numpy_T%d_B%d' % (Tq, B)] = {'seconds': s, 'elements_per_s': Tq * B / s,
                                                 'GBps': Tq * B * 28 / s / 1e9, 'kind': 'port', 'cores': 1,
                                                 'what': 'vtrace.py:99-137 per-t loop in numpy'}
    return out
