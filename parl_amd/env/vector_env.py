"""parl.env.vector_env.VectorEnv with the reference's contract (parl/env/vector_env.py:26-63):
`VectorEnv(envs)`, `reset() -> [obs]`, `step(actions) -> ([obs], [reward], [done], [info])` with
auto-reset, lists of numpy arrays / Python scalars on the HOST — for scripts written against the
reference (benchmark/torch/a2c/actor.py:30-101, examples/IMPALA/actor.py:29-91).  The envs are the
handles `wrap_deepmind(gym.make(id), dim, obs_format)` returns; they all run in ONE DeviceVectorEnv
(one kernel launch per step for the whole list).  Every step costs a device-to-host copy of the
observations; the device-native loop (parl_amd.rollout) has none — this class is the drop-in
boundary, not the fast path."""
import itertools

import numpy as np
import torch

from .atari_wrappers import WrappedDeviceAtariEnv
from .device_vector_env import DeviceVectorEnv

__all__ = ['VectorEnv']

_next_env_id = itertools.count()  # distinct RNG streams (noop counts) for every env of the process


class VectorEnv(object):
    def __init__(self, envs, seed=0, device=None):
        if not envs or not all(isinstance(e, WrappedDeviceAtariEnv) for e in envs):
            raise TypeError('VectorEnv: expected a list of envs made by wrap_deepmind(gym.make(...)) of the device path')
        first = envs[0]
        if any((e.env_id, e.dim, e.obs_format) != (first.env_id, first.dim, first.obs_format) for e in envs):
            raise ValueError('VectorEnv: all envs must share env id, dim and obs_format')
        self.envs = envs
        self.envs_num = len(envs)
        id0 = next(_next_env_id)
        for _ in range(self.envs_num - 1):
            next(_next_env_id)
        self.dev_env = DeviceVectorEnv(first.env_id, self.envs_num, dim=first.dim, horizon=64, seed=seed,
                                       env_id0=id0, device=device)
        self._nhwc = first.obs_format == 'NHWC'
        for i, e in enumerate(envs):  # MonitorEnv.get_total_steps also counts the running episode (atari_wrappers.py:73-77)
            e.monitor._running_steps = lambda i=i: self.dev_env.running_episode_steps()[i].item()

    def _obs_list(self, obs):
        a = obs.cpu().numpy()
        if self._nhwc:
            a = a.transpose(0, 2, 3, 1)
        return list(a)

    def reset(self):
        """vector_env.py:34-39"""
        obs = self._obs_list(self.dev_env.reset())
        for e in self.envs:
            if e.test_env is not None:  # TestEnv.reset ran for every env
                e.test_env._on_reset()
        return obs

    def step(self, actions):
        """vector_env.py:41-63 (the obs returned for a done env is its reset obs)"""
        a = torch.as_tensor(np.asarray(actions).reshape(-1), dtype=torch.int64, device=self.dev_env.device)
        obs, rew, done, info = self.dev_env.step(a)
        ret = info['episode_returns'].cpu().numpy()
        ln = info['episode_lengths'].cpu().numpy()
        dones = [bool(x) for x in done.cpu().numpy()]
        infos = []
        for i, e in enumerate(self.envs):
            if ln[i] > 0:  # MonitorEnv closed an episode of env i in this step
                e.monitor._push(ret[i], ln[i])
                infos.append({'episode': {'r': float(ret[i]), 'l': int(ln[i])}})
            else:
                infos.append({})
            if dones[i] and e.test_env is not None:  # the auto-reset of a done env went through TestEnv.reset (:55-57)
                e.test_env._on_reset()
        self.dev_env.check_faults()
        return self._obs_list(obs), [float(x) for x in rew.cpu().numpy()], dones, infos
