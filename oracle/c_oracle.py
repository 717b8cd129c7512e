"""ctypes loader for oracle/libparl_oracle.so (the plain-C restatement) with numpy in/out.

TEST INFRASTRUCTURE ONLY — see oracle/__init__.py."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, 'libparl_oracle.so')
_lib = None


def build():
    subprocess.check_call(['make', '-C', _HERE, '-s'])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = ctypes.CDLL(_LIB)
        _lib.oracle_philox_uniform53.restype = ctypes.c_double
        _lib.oracle_philox_uniform53.argtypes = [ctypes.c_uint64] * 3
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


_NAN = float('nan')


def _thr(x):
    return _NAN if x is None else float(x)


def vtrace(blp, tlp, discounts, rewards, values, bootstrap, clip_rho=1.0, clip_pg_rho=1.0):
    blp, tlp, discounts, rewards, values = [_c(x, np.float32) for x in (blp, tlp, discounts, rewards, values)]
    bootstrap = _c(bootstrap, np.float32)
    T, B = blp.shape
    vs = np.empty((T, B), np.float32)
    pg = np.empty((T, B), np.float32)
    rc = lib().oracle_vtrace_f32(_p(blp), _p(tlp), _p(discounts), _p(rewards), _p(values), _p(bootstrap),
                                 _p(vs), _p(pg), T, B, ctypes.c_float(_thr(clip_rho)),
                                 ctypes.c_float(_thr(clip_pg_rho)))
    assert rc == 0
    return vs, pg


def vtrace_from_logits(blogits, tlogits, actions, rewards, dones, values, gamma, clip_rho=1.0,
                       clip_pg_rho=1.0, time_major=True):
    blogits, tlogits = _c(blogits, np.float32), _c(tlogits, np.float32)
    actions = _c(actions, np.int64)
    rewards, values = _c(rewards, np.float32), _c(values, np.float32)
    dones = _c(dones, np.uint8)
    if time_major:
        T, B, A = tlogits.shape
        osh = (T - 1, B)
    else:
        B, T, A = tlogits.shape
        osh = (B, T - 1)
    vs, pg, tlp, blp = [np.zeros(osh, np.float32) for _ in range(4)]
    rc = lib().oracle_vtrace_from_logits_f32(_p(blogits), _p(tlogits), _p(actions), _p(rewards), _p(dones),
                                             _p(values), _p(vs), _p(pg), _p(tlp), _p(blp), T, B, A,
                                             1 if time_major else 0, ctypes.c_float(gamma),
                                             ctypes.c_float(_thr(clip_rho)), ctypes.c_float(_thr(clip_pg_rho)))
    assert rc == 0
    return vs, pg, tlp, blp


def gae(rewards, values, dones, next_value, gamma, lam, last_done=None, done_convention=0, accum_f64=False):
    rewards, values = _c(rewards, np.float32), _c(values, np.float32)
    next_value = _c(next_value, np.float32).reshape(-1)
    dones = np.ascontiguousarray(dones)
    is_f32 = dones.dtype == np.float32
    if not is_f32:
        dones = _c(dones, np.uint8)
    if last_done is not None:
        last_done = _c(last_done, dones.dtype).reshape(-1)
    T, B = rewards.shape
    adv = np.empty((T, B), np.float32)
    ret = np.empty((T, B), np.float32)
    rc = lib().oracle_gae_f32(_p(rewards), _p(values), _p(dones), _p(next_value), _p(last_done), _p(adv), _p(ret),
                              T, B, ctypes.c_float(gamma), ctypes.c_float(lam), int(done_convention),
                              1 if is_f32 else 0, 1 if accum_f64 else 0)
    assert rc == 0
    return adv, ret


def discount_cumsum(x, gamma, dones=None, accum_f64=False):
    x = _c(x, np.float32)
    T, B = x.shape
    if dones is not None:
        dones = _c(dones, np.uint8)
    out = np.empty((T, B), np.float32)
    rc = lib().oracle_discount_cumsum_f32(_p(x), _p(dones), _p(out), T, B, ctypes.c_float(gamma),
                                          1 if accum_f64 else 0)
    assert rc == 0
    return out


def adv_normalize(adv, idx=None, eps=1e-8):
    adv = _c(adv, np.float32).reshape(-1)
    if idx is not None:
        idx = _c(idx, np.int64).reshape(-1)
        n = idx.size
    else:
        n = adv.size
    out = np.empty(n, np.float32)
    ms = np.empty(2, np.float32)
    rc = lib().oracle_adv_normalize_f32(_p(adv), _p(idx), _p(out), ctypes.c_int64(n), ctypes.c_float(eps), _p(ms))
    assert rc == 0
    return out, ms


def categorical_sample(probs, uniforms):
    probs = _c(probs, np.float32)
    uniforms = _c(uniforms, np.float64)
    B, A = probs.shape
    actions = np.empty(B, np.int64)
    rc = lib().oracle_categorical_sample_f32(_p(probs), _p(uniforms), _p(actions), B, A)
    assert rc == 0
    return actions


def philox_uniform53(seed, offset, row):
    return lib().oracle_philox_uniform53(seed, offset, row)


def policy_sample(x, seed, offset, row0=0, is_logits=True):
    x = _c(x, np.float32)
    B, A = x.shape
    actions = np.empty(B, np.int64)
    probs = np.empty((B, A), np.float32)
    uni = np.empty(B, np.float64)
    rc = lib().oracle_policy_sample_f32(_p(x), 1 if is_logits else 0, _p(actions), _p(probs), _p(uni), B, A,
                                        ctypes.c_uint64(seed), ctypes.c_uint64(offset), ctypes.c_uint64(row0))
    assert rc == 0
    return actions, probs, uni


# ---------------------------------------------------------------------------------------
# Atari: emulator + wrapper chain + VectorEnv (atari_oracle.c, atari_env_oracle.c)
# ---------------------------------------------------------------------------------------
GAME_IDS = {'pong': 1, 'breakout': 2}


def frame_tables(dim):
    L = lib()
    L.oracle_frame_tables_bytes.restype = ctypes.c_size_t
    n = L.oracle_frame_tables_bytes(int(dim))
    blob = np.zeros(n, np.uint8)
    L.oracle_frame_tables_init(_p(blob), int(dim))
    return blob


def frame_post(frames0, frames1, dim, fmt):
    """max (optional) + RGB2GRAY + INTER_AREA resize; frames [E,210,160(,3)] u8 -> [E,dim,dim]"""
    f0 = _c(frames0, np.uint8)
    f1 = None if frames1 is None else _c(frames1, np.uint8)
    E = f0.shape[0]
    out = np.zeros((E, dim, dim), np.uint8)
    blob = frame_tables(dim)
    rc = lib().oracle_frame_post_u8(_p(f0), _p(f1), int(fmt), _p(out), ctypes.c_int64(dim * dim), E, int(dim),
                                    _p(blob))
    assert rc == 0
    return out


class VecEnv:
    """VectorEnv([wrap_deepmind(gym.make(...), dim, obs_format='NCHW')] * E) restated on CPU."""

    def __init__(self, rom_bytes, game, E, dim=84, seed=0, env_id0=0, max_episode_steps=400000):
        L = lib()
        L.oracle_vec_new.restype = ctypes.c_void_p
        self.L, self.E, self.dim = L, E, dim
        self.h = ctypes.c_void_p(
            L.oracle_vec_new(rom_bytes, len(rom_bytes), GAME_IDS[game], E, dim, ctypes.c_uint64(seed),
                             ctypes.c_uint64(env_id0), ctypes.c_int64(max_episode_steps)))
        self.num_actions = L.oracle_vec_num_actions(self.h)

    def reset(self):
        obs = np.zeros((self.E, 4, self.dim, self.dim), np.uint8)
        self.L.oracle_vec_reset(self.h, _p(obs))
        return obs

    def step(self, actions):
        actions = _c(actions, np.int64)
        obs = np.zeros((self.E, 4, self.dim, self.dim), np.uint8)
        rew = np.zeros(self.E, np.float32)
        done = np.zeros(self.E, np.uint8)
        self.L.oracle_vec_step(self.h, _p(actions), _p(obs), _p(rew), _p(done))
        return obs, rew, done

    def pop_episodes(self, env):
        r = np.zeros(64, np.float64)
        n_ = np.zeros(64, np.int64)
        n = self.L.oracle_vec_pop_episodes(self.h, env, _p(r), _p(n_), 64)
        return list(zip(r[:n].tolist(), n_[:n].tolist()))

    def ram(self, env):
        out = np.zeros(128, np.uint8)
        self.L.oracle_vec_ram(self.h, env, _p(out))
        return out

    def raw_frames(self, env):
        out = np.zeros((2, 210, 160), np.uint8)
        self.L.oracle_vec_raw_frames(self.h, env, _p(out))
        return out

    def lives(self, env):
        return self.L.oracle_vec_lives(self.h, env)

    def total_steps(self, env):
        """MonitorEnv.get_total_steps() of env (parl/env/atari_wrappers.py:73-77,90-91): every raw step so far"""
        self.L.oracle_vec_total_steps.restype = ctypes.c_int64
        return int(self.L.oracle_vec_total_steps(self.h, env))

    def __del__(self):
        try:
            self.L.oracle_vec_free(self.h)
        except Exception:
            pass


# ---- PPO: VecNormalizeEnv running statistics, minibatch gather (ppo_oracle.c) ----
class VecNormalize(object):
    """E independent VecNormalizeEnv instances (parl/env/mujoco_wrappers.py:95-169) as arrays"""

    def __init__(self, E, D, gamma=0.99, clipob=10.0, cliprew=10.0, epsilon=1e-8, rms_epsilon=1e-4):
        self.E, self.D = E, D
        self.gamma, self.clipob, self.cliprew, self.eps = gamma, clipob, cliprew, epsilon
        self.ob_mean = np.zeros((E, D), np.float64)
        self.ob_var = np.ones((E, D), np.float64)
        self.ob_count = np.full(E, rms_epsilon, np.float64)
        self.ret_mean = np.zeros(E, np.float64)
        self.ret_var = np.ones(E, np.float64)
        self.ret_count = np.full(E, rms_epsilon, np.float64)
        self.ret = np.zeros(E, np.float64)
        self.training = True

    def filter_obs(self, raw, mask=None, out=None):
        raw = _c(raw, np.float64)
        if out is None:
            out = np.zeros((self.E, self.D), np.float64)
        m = None if mask is None else _c(mask, np.uint8)
        rc = lib().oracle_vecnorm_obs_f64(_p(raw), _p(self.ob_mean), _p(self.ob_var), _p(self.ob_count), _p(m),
                                          None, _p(out), self.E, self.D, ctypes.c_double(self.clipob),
                                          ctypes.c_double(self.eps), 1 if self.training else 0)
        assert rc == 0
        return out

    def filter_reward(self, rew, done):
        rew, done = _c(rew, np.float64), _c(done, np.uint8)
        out = np.zeros(self.E, np.float64)
        rc = lib().oracle_vecnorm_reward_f64(_p(rew), _p(done), _p(self.ret), _p(self.ret_mean), _p(self.ret_var),
                                             _p(self.ret_count), None, _p(out), self.E, ctypes.c_double(self.gamma),
                                             ctypes.c_double(self.cliprew), ctypes.c_double(self.eps))
        assert rc == 0
        return out


def ppo_sample_batch(obs, actions, logprobs, advantages, returns, values, idx):
    """flattened rollout arrays ([N, ...]) gathered by idx -> (obs [M,Do], actions [M,Da], 4 x [M])"""
    N = logprobs.size
    obs2, act2 = _c(obs, np.float32).reshape(N, -1), _c(actions, np.float32).reshape(N, -1)
    idx = _c(idx, np.int64)
    M = idx.size
    outs = [np.empty((M, obs2.shape[1]), np.float32), np.empty((M, act2.shape[1]), np.float32)] + \
           [np.empty(M, np.float32) for _ in range(4)]
    flat = [_c(x, np.float32).reshape(-1) for x in (logprobs, advantages, returns, values)]
    rc = lib().oracle_ppo_sample_batch_f32(_p(obs2), _p(act2), *[_p(x) for x in flat], _p(idx),
                                           *[_p(o) for o in outs], ctypes.c_int64(N), ctypes.c_int64(M),
                                           obs2.shape[1], act2.shape[1])
    assert rc == 0
    return outs
