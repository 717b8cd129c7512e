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
