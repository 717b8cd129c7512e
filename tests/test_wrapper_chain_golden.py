"""The env wrapper chain against fixtures produced by the REFERENCE's own Python wrappers
(parl/env/atari_wrappers.py, compat_wrappers.py, vector_env.py executed from /root/reference by
tests/golden/make_wrapper_golden.py, on the oracle's emulator/image primitives):
  * CPU: the C oracle's flattened chain (oracle/atari_env_oracle.c) reproduces every reward, done,
    observation (CRC32 of each stacked obs, a few stored in full) and MonitorEnv episode record;
  * GPU (-m gpu): the HIP env kernel + frame_post + frame-stack ring do the same through the C ABI.
"""
import os
import zlib

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

CASES = ['pong_84', 'pong_42_timelimit', 'breakout_84', 'breakout_42_timelimit']
GYM_ID = {'pong': 'PongNoFrameskip-v4', 'breakout': 'BreakoutNoFrameskip-v4'}


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xffffffff


def load(case):
    z = np.load(os.path.join(GOLDEN, 'wrapper_chain_%s.npz' % case))
    return {k: z[k] for k in z.files}


def rom(game):
    from parl_amd.env import find_rom
    try:
        return find_rom(game)
    except FileNotFoundError:
        pytest.skip('cartridge %s.bin not present (roms/ is user-supplied data)' % game)


@pytest.mark.parametrize('case', CASES)
def test_oracle_chain_matches_reference_wrappers(oracle, case):
    g = load(case)
    game, E, dim = str(g['game']), int(g['E']), int(g['dim'])
    v = oracle.VecEnv(rom(game), game, E, dim, seed=int(g['seed']), max_episode_steps=int(g['max_episode_steps']))
    obs = v.reset()
    assert np.array_equal(obs, g['reset_obs'])
    full = {int(t): g['full_obs'][i] for i, t in enumerate(g['full_steps'])}
    eps = []
    for t in range(g['actions'].shape[0]):
        o, r, d = v.step(g['actions'][t])
        assert np.array_equal(r, g['rewards'][t]), 'reward mismatch at step %d' % t
        assert np.array_equal(d, g['dones'][t]), 'done mismatch at step %d' % t
        assert [crc(x) for x in o] == list(g['obs_crc'][t]), 'obs mismatch at step %d' % t
        if t in full:
            assert np.array_equal(o, full[t])
        for e in range(E):
            eps += [(e, r_, l_) for r_, l_ in v.pop_episodes(e)]
    ref = sorted(map(tuple, g['episodes'].tolist()))
    assert sorted((float(e), float(r_), float(l_)) for e, r_, l_ in eps) == ref


@pytest.mark.gpu
@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('cache', [True, False])
def test_device_env_matches_reference_wrappers(dev, case, cache):
    import torch
    from parl_amd.env import DeviceVectorEnv
    g = load(case)
    game, E, dim = str(g['game']), int(g['E']), int(g['dim'])
    env = DeviceVectorEnv(GYM_ID[game], E, dim=dim, horizon=16, seed=int(g['seed']), device=dev, rom_bytes=rom(game),
                          max_episode_steps=int(g['max_episode_steps']), use_reset_cache=cache)
    assert np.array_equal(env.reset().cpu().numpy(), g['reset_obs'])
    full = {int(t): g['full_obs'][i] for i, t in enumerate(g['full_steps'])}
    eps = []
    for t in range(g['actions'].shape[0]):
        o, r, d, info = env.step(torch.from_numpy(g['actions'][t]).to(dev))
        o = o.cpu().numpy()
        assert np.array_equal(r.cpu().numpy(), g['rewards'][t]), 'reward mismatch at step %d' % t
        assert np.array_equal(d.cpu().numpy().astype(np.uint8), g['dones'][t]), 'done mismatch at step %d' % t
        assert [crc(x) for x in o] == list(g['obs_crc'][t]), 'obs mismatch at step %d' % t
        if t in full:
            assert np.array_equal(o, full[t])
        ln = info['episode_lengths'].cpu().numpy()
        rt = info['episode_returns'].cpu().numpy()
        eps += [(float(e), float(rt[e]), float(ln[e])) for e in range(E) if ln[e] > 0]
    env.check_faults()
    # the device reports at most ONE closed episode per env-step (the last); the fixtures were
    # chosen so that no step closes two
    assert sorted(eps) == sorted(map(tuple, g['episodes'].tolist()))


# ---- wrap_deepmind(test=True): TestEnv (parl/env/atari_wrappers.py:309-353) -------------------------------------
def _check_test_rows(g, t, envs):
    k = int(g['test_episodes'])
    for e, env in enumerate(envs):
        assert bool(env.get_real_done()) == bool(g['real_done'][t, e]), 'get_real_done, row %d env %d' % (t, e)
        want = g['eval_rewards'][t, e]
        got = env.get_eval_rewards()
        if np.isnan(want[0]):
            assert got is None, (t, e, got)
        else:
            assert list(got) == list(want[:k]), (t, e, got, want)


def test_testenv_bookkeeping_matches_the_reference_class(oracle):
    """parl_amd.env.atari_wrappers.TestEnv on MonitorEnv records delivered the way VectorEnv delivers them, the env
    stream from the C oracle: get_real_done() / get_eval_rewards() after the reset and after every step equal what
    the reference's TestEnv reported on the same trajectory (fixture made by its own code)."""
    from parl_amd.env.atari_wrappers import DeviceAtariEnv, TestEnv, WrappedDeviceAtariEnv, get_wrapper_by_cls
    g = load('breakout_42_test')
    game, E, dim, k = str(g['game']), int(g['E']), int(g['dim']), int(g['test_episodes'])
    v = oracle.VecEnv(rom(game), game, E, dim, seed=int(g['seed']), max_episode_steps=int(g['max_episode_steps']))

    class Handle(WrappedDeviceAtariEnv):   # a handle without the device library behind it
        def __init__(self):
            self.monitor = __import__('parl_amd.env.atari_wrappers', fromlist=['MonitorEnv']).MonitorEnv()
            self.test_env = TestEnv(self.monitor, k)

    envs = [Handle() for _ in range(E)]
    assert isinstance(get_wrapper_by_cls(envs[0], TestEnv), TestEnv) and DeviceAtariEnv is not None
    assert np.array_equal(v.reset(), g['reset_obs'])
    for env in envs:
        env.test_env._on_reset()
    _check_test_rows(g, 0, envs)
    n_real = 0
    for t in range(g['actions'].shape[0]):
        o, r, d = v.step(g['actions'][t])
        assert np.array_equal(d, g['dones'][t]) and np.array_equal(r, g['rewards'][t])
        for e, env in enumerate(envs):
            for ret, ln in v.pop_episodes(e):
                env.monitor._push(ret, ln)
            if d[e]:
                env.test_env._on_reset()
        _check_test_rows(g, t + 1, envs)
        n_real += int(g['real_done'][t + 1].sum())
    assert n_real > 0 and g['dones'].sum() > len(g['episodes'])   # windows closed; life-loss dones stayed dones


@pytest.mark.gpu
def test_device_vector_env_with_test_wrapper_matches_reference(dev, monkeypatch):
    """the drop-in form on the device: wrap_deepmind(gym.make(id), dim, obs_format, test=True, test_episodes=2) handles
    in parl.env.vector_env.VectorEnv — observations, rewards, dones AND TestEnv's two getters against the fixture"""
    import itertools
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'compat'))
    try:
        for m in ('gym', 'parl'):
            sys.modules.pop(m, None)
        import gym
        from parl.env.atari_wrappers import wrap_deepmind, TestEnv, get_wrapper_by_cls
        from parl.env import vector_env as ve
        g = load('breakout_42_test')
        game, E, dim, k = str(g['game']), int(g['E']), int(g['dim']), int(g['test_episodes'])
        rom(game)
        monkeypatch.setattr(ve, '_next_env_id', itertools.count())   # env ids 0 .. E-1 as in the fixture
        envs = [wrap_deepmind(gym.make(GYM_ID[game]), dim=dim, obs_format='NCHW', test=True, test_episodes=k) for _ in range(E)]
        assert isinstance(get_wrapper_by_cls(envs[0], TestEnv), TestEnv)
        vec = ve.VectorEnv(envs, seed=int(g['seed']), device=dev)
        obs = vec.reset()
        assert np.array_equal(np.stack(obs), g['reset_obs'])
        _check_test_rows(g, 0, envs)
        for t in range(g['actions'].shape[0]):
            o, r, d, info = vec.step(g['actions'][t])
            assert r == [float(x) for x in g['rewards'][t]] and d == [bool(x) for x in g['dones'][t]], t
            assert [crc(x) for x in o] == list(g['obs_crc'][t]), 'obs mismatch at step %d' % t
            _check_test_rows(g, t + 1, envs)
    finally:
        sys.path.remove(os.path.join(ROOT, 'compat'))
