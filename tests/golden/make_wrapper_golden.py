"""Generate tests/golden/wrapper_chain_*.npz by running the REFERENCE's own wrapper code.

Build-container only (reads /root/reference).  What is real and what is stubbed:

  REAL (reference source, executed as-is from /root/reference):
    parl/env/atari_wrappers.py   wrap_deepmind chain: MonitorEnv, NoopResetEnv, MaxAndSkipEnv,
                                 EpisodicLifeEnv, FireResetEnv, WarpFrame, ClipRewardEnv, FrameStack
    parl/env/compat_wrappers.py  CompatWrapper (the never-reset step counter quirk)
    parl/env/vector_env.py       VectorEnv.reset / step with auto-reset
  STUBBED (third party, absent from this container; SURVEY.md §8c "parity unpinned"):
    gym      Wrapper / ObservationWrapper / RewardWrapper / spaces.Box base classes with gym 0.12.1's
             published behaviour (attribute delegation, `unwrapped`, `spec`), plus TimeLimit
             (`_max_episode_steps`, `_elapsed_steps`) restated below;
    gym AtariEnv + ALE   served by the CPU oracle's ALE layer (oracle/atari_oracle.c) — one
             ale.act() per step (frameskip 1), getScreenRGB through the NTSC palette;
    cv2      cvtColor(RGB2GRAY) / resize(INTER_AREA) served by the oracle's restatement of
             OpenCV (oracle/frame_oracle.c);
    np_random.randint(1, 31)  served by the counter-based stream the device env uses
             (philox4x32-10(seed; reset_count, env_id)[0] % 30 + 1) so that trajectories are
             reproducible (the reference never seeds).

So these fixtures pin the WRAPPER STATE MACHINE (who resets when, which frames are maxed, how
lives/done/reward/monitor statistics flow, frame-stack contents, auto-reset) of our C oracle
(oracle/atari_env_oracle.c) and of the HIP env kernel against the reference's Python, on identical
emulator + image primitives.

    python tests/golden/make_wrapper_golden.py
"""
import ctypes
import importlib.util
import os
import sys
import types
import zlib

import numpy as np

sys.dont_write_bytecode = True  # modules are imported from /root/reference by path: never leave a __pycache__ there

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = '/root/reference'

from oracle import c_oracle  # noqa: E402

# --------------------------------------------------------------------------------------------
# gym stub (gym 0.12.1 semantics of the pieces the wrappers use)
# --------------------------------------------------------------------------------------------
gym = types.ModuleType('gym')
gym.__version__ = '0.12.1'


class _Env(object):
    @property
    def unwrapped(self):
        return self


class _Wrapper(_Env):
    def __init__(self, env):
        self.env = env
        self.action_space = getattr(env, 'action_space', None)
        self.observation_space = getattr(env, 'observation_space', None)

    def __getattr__(self, name):  # gym.Wrapper.__getattr__: delegate public attributes
        if name.startswith('__'):
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env.unwrapped

    @property
    def spec(self):
        return self.env.spec

    def step(self, action):
        return self.env.step(action)

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)


class _ObservationWrapper(_Wrapper):
    def reset(self, **kwargs):
        return self.observation(self.env.reset(**kwargs))

    def step(self, action):
        o, r, d, i = self.env.step(action)
        return self.observation(o), r, d, i


class _RewardWrapper(_Wrapper):
    def step(self, action):
        o, r, d, i = self.env.step(action)
        return o, self.reward(r), d, i


class _Box(object):
    def __init__(self, low, high, shape, dtype):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype


gym.Env, gym.Wrapper, gym.ObservationWrapper, gym.RewardWrapper = _Env, _Wrapper, _ObservationWrapper, _RewardWrapper
spaces = types.ModuleType('gym.spaces')
spaces.Box = _Box
gym.spaces = spaces

# --------------------------------------------------------------------------------------------
# cv2 stub: the two calls of WarpFrame.observation, served by the oracle's OpenCV restatement
# --------------------------------------------------------------------------------------------
cv2 = types.ModuleType('cv2')
cv2.ocl = types.SimpleNamespace(setUseOpenCL=lambda flag: None)
cv2.COLOR_RGB2GRAY = 7
cv2.INTER_AREA = 3
_stash = {}


def _cvtColor(frame, code):
    assert code == cv2.COLOR_RGB2GRAY and frame.shape == (210, 160, 3) and frame.dtype == np.uint8
    f = frame.astype(np.uint32)
    gray = ((f[..., 0] * 4899 + f[..., 1] * 9617 + f[..., 2] * 1868 + 8192) >> 14).astype(np.uint8)
    _stash['rgb'], _stash['gray'] = np.ascontiguousarray(frame), gray
    return gray


def _resize(frame, size, interpolation=None):
    assert interpolation == cv2.INTER_AREA and size[0] == size[1]
    assert frame is _stash['gray']  # WarpFrame feeds cvtColor's output straight into resize
    return c_oracle.frame_post(_stash['rgb'][None], None, size[0], 0)[0]


cv2.cvtColor, cv2.resize = _cvtColor, _resize


def load_reference_env_modules():
    sys.modules['gym'] = gym
    sys.modules['gym.spaces'] = spaces
    sys.modules['cv2'] = cv2
    for name in ('parl', 'parl.env'):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    out = {}
    for mod in ('compat_wrappers', 'atari_wrappers', 'vector_env'):
        full = 'parl.env.' + mod
        spec = importlib.util.spec_from_file_location(full, os.path.join(REF, 'parl', 'env', mod + '.py'))
        m = importlib.util.module_from_spec(spec)
        sys.modules[full] = m
        spec.loader.exec_module(m)
        out[mod] = m
    return out


# --------------------------------------------------------------------------------------------
# gym.make('<Game>NoFrameskip-v4') stand-in: AtariEnv(frameskip=1) on the oracle's ALE layer,
# inside gym's TimeLimit
# --------------------------------------------------------------------------------------------
ALE_ACTIONS = {'pong': [0, 1, 3, 4, 11, 12], 'breakout': [0, 1, 3, 4]}  # ALE minimal action sets
MEANINGS = {0: 'NOOP', 1: 'FIRE', 3: 'RIGHT', 4: 'LEFT', 11: 'RIGHTFIRE', 12: 'LEFTFIRE'}


class _NpRandom(object):
    """`env.unwrapped.np_random` with the device env's counter-based noop stream."""

    def __init__(self, seed, env_id):
        self.seed, self.env_id, self.count = seed, env_id, 0

    def randint(self, lo, hi):
        assert (lo, hi) == (1, 31)
        w = (ctypes.c_uint32 * 4)()
        c_oracle.lib().oracle_philox4x32_10(ctypes.c_uint64(self.seed), ctypes.c_uint64(self.count),
                                            ctypes.c_uint64(self.env_id), w)
        self.count += 1
        return 1 + int(w[0]) % 30


class OracleAtariEnv(_Env):
    def __init__(self, game, seed, env_id):
        L = c_oracle.lib()
        L.oracle_ale_new.restype = ctypes.c_void_p
        rom = open(os.path.join(REF, 'benchmark/fluid/DQN_variant/rom_files', game + '.bin'), 'rb').read()
        self.L = L
        self.h = ctypes.c_void_p(L.oracle_ale_new(rom, len(rom), c_oracle.GAME_IDS[game]))
        self.game = game
        self.spec = types.SimpleNamespace(id={'pong': 'Pong', 'breakout': 'Breakout'}[game] + 'NoFrameskip-v4')
        self.observation_space = _Box(0, 255, (210, 160, 3), np.uint8)
        self._action_set = ALE_ACTIONS[game]
        self.np_random = _NpRandom(seed, env_id)
        self.ale = types.SimpleNamespace(lives=lambda: self.L.oracle_ale_lives(self.h))
        pal = (ctypes.c_uint32 * 128)()
        L.oracle_palette(pal)
        p = np.frombuffer(pal, np.uint32)
        self.pal = np.stack([(p >> 16) & 255, (p >> 8) & 255, p & 255], -1).astype(np.uint8)
        self.fb = np.zeros((210, 160), np.uint8)

    def get_action_meanings(self):
        return [MEANINGS[a] for a in self._action_set]

    def _rgb(self):
        return self.pal[self.fb >> 1]  # getScreenRGB

    def reset(self):
        self.L.oracle_ale_reset(self.h, self.fb.ctypes.data_as(ctypes.c_void_p))
        return self._rgb()

    def step(self, a):
        r = self.L.oracle_ale_act(self.h, self._action_set[a], self.fb.ctypes.data_as(ctypes.c_void_p))
        return self._rgb(), float(r), bool(self.L.oracle_ale_terminal(self.h)), {'ale.lives': self.ale.lives()}


class TimeLimit(_Wrapper):
    """gym 0.12.1 wrappers/time_limit.py: done once _elapsed_steps >= _max_episode_steps."""

    def __init__(self, env, max_episode_steps):
        _Wrapper.__init__(self, env)
        self._max_episode_steps = max_episode_steps
        self._elapsed_steps = 0

    def step(self, action):
        o, r, d, i = self.env.step(action)
        self._elapsed_steps += 1
        if self._elapsed_steps >= self._max_episode_steps:
            d = True
        return o, r, d, i

    def reset(self, **kwargs):
        self._elapsed_steps = 0
        return self.env.reset(**kwargs)


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xffffffff


def run_case(mods, game, E, dim, seed, steps, max_episode_steps, action_seed, keep_full, test_episodes=None):
    aw, ve = mods['atari_wrappers'], mods['vector_env']
    envs = []
    for e in range(E):
        base = TimeLimit(OracleAtariEnv(game, seed, e), max_episode_steps)
        if test_episodes:  # the evaluation wrapper on top of the chain (atari_wrappers.py:309-353, :383-384)
            envs.append(aw.wrap_deepmind(base, dim=dim, obs_format='NCHW', test=True, test_episodes=test_episodes))
        else:
            envs.append(aw.wrap_deepmind(base, dim=dim, obs_format='NCHW'))
    vec = ve.VectorEnv(envs)
    obs = vec.reset()
    A = len(ALE_ACTIONS[game])
    rng = np.random.default_rng(action_seed)
    actions = rng.integers(0, A, (steps, E)).astype(np.int64)
    out = {
        'game': game, 'E': E, 'dim': dim, 'seed': seed, 'max_episode_steps': max_episode_steps, 'actions': actions,
        'reset_obs_crc': np.array([crc(o) for o in obs], np.uint32), 'reset_obs': np.stack(obs),
    }
    rew = np.zeros((steps, E), np.float32)
    done = np.zeros((steps, E), np.uint8)
    ocrc = np.zeros((steps, E), np.uint32)
    k = test_episodes or 0
    real_done = np.zeros((steps + 1, E), np.uint8)           # TestEnv.get_real_done() after reset / every step
    eval_rew = np.full((steps + 1, E, max(k, 1)), np.nan)    # TestEnv.get_eval_rewards() (NaN: None)

    def record_test(row):
        for e, env in enumerate(envs):
            real_done[row, e] = env.get_real_done()
            ev = env.get_eval_rewards()
            if ev is not None:
                eval_rew[row, e] = ev

    if k:
        record_test(0)
    full = {}
    for t in range(steps):
        o, r, d, _ = vec.step(actions[t])
        rew[t], done[t] = r, d
        ocrc[t] = [crc(x) for x in o]
        if k:
            record_test(t + 1)
        if t in keep_full:
            full[t] = np.stack(o)
    out.update(rewards=rew, dones=done, obs_crc=ocrc)
    out['full_steps'] = np.array(sorted(full), np.int64)
    out['full_obs'] = np.stack([full[t] for t in sorted(full)]) if full else np.zeros((0, ))
    # MonitorEnv statistics: (unclipped return, length in raw frames) of every closed episode
    eps = []
    for e, env in enumerate(envs):
        mon = aw.get_wrapper_by_cls(env, aw.MonitorEnv)
        for r_, l_ in mon.next_episode_results():
            eps.append((e, r_, l_))
    out['episodes'] = np.array(eps, np.float64).reshape(-1, 3)
    if k:
        out.update(test_episodes=k, real_done=real_done, eval_rewards=eval_rew)
    return out


CASES = [
    # name, game, E, dim, seed, steps, max_episode_steps, action_seed
    ('pong_84', 'pong', 2, 84, 3, 300, 400000, 0),
    ('pong_42_timelimit', 'pong', 2, 42, 11, 420, 1000, 1),  # TimeLimit + CompatWrapper dones
    ('breakout_84', 'breakout', 2, 84, 5, 500, 400000, 2),  # life loss, FIRE reset, game over
    ('breakout_42_timelimit', 'breakout', 2, 42, 7, 400, 700, 3),
]

# wrap_deepmind(test=True): TestEnv's evaluation bookkeeping on top of the same chain
TEST_CASES = [('breakout_42_test', 'breakout', 2, 42, 9, 1100, 400000, 4, 2)]

if __name__ == '__main__':
    mods = load_reference_env_modules()
    for name, game, E, dim, seed, steps, mes, aseed, k in TEST_CASES:
        res = run_case(mods, game, E, dim, seed, steps, mes, aseed, keep_full={0, steps - 1}, test_episodes=k)
        path = os.path.join(HERE, 'wrapper_chain_%s.npz' % name)
        np.savez_compressed(path, **res)
        print(name, 'dones', int(res['dones'].sum()), 'episodes', len(res['episodes']), 'real_done rows',
              int(res['real_done'].sum()), '->', os.path.getsize(path), 'bytes')
    if os.environ.get('ONLY_TEST_CASES'):
        sys.exit(0)
    for name, game, E, dim, seed, steps, mes, aseed in CASES:
        res = run_case(mods, game, E, dim, seed, steps, mes, aseed, keep_full={0, steps // 2, steps - 1})
        path = os.path.join(HERE, 'wrapper_chain_%s.npz' % name)
        np.savez_compressed(path, **res)
        print(name, 'dones', int(res['dones'].sum()), 'episodes', len(res['episodes']), 'abs reward',
              float(np.abs(res['rewards']).sum()), '->', os.path.getsize(path), 'bytes')
