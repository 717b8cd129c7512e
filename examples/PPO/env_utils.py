"""ParallelEnv of examples/PPO/env_utils.py:28-115 for the two paths of this repository:

* discrete (Atari): the environments live on the GPU (parl_amd.env.DeviceVectorEnv, the same
  emulator + wrap_deepmind chain IMPALA / A2C use), observations never leave HBM;
* continuous (MuJoCo, BASELINE configs[4]): the simulators step on the host; their raw float64
  observations / rewards / dones are uploaded once per step and normalised on the GPU by
  parl_amd.env.DeviceVecNormalize (one VecNormalizeEnv per env, bit-identical statistics).
  gym / mujoco-py are not installed in this image, so `make_host_envs` falls back to a synthetic
  linear-dynamics simulator with the HalfCheetah shapes (obs 17, act 6, 1000-step episodes) —
  a stand-in for the physics only; everything downstream of `env.step` is the real path."""
import collections

import numpy as np
import torch

from parl_amd.env import DeviceVecNormalize, DeviceVectorEnv

Space = collections.namedtuple('Space', ['shape', 'n'])
GAMMA = 0.99


class SyntheticHostSim(object):
    """stand-in for E host-stepped MuJoCo simulators (float64 observations like mujoco-py)"""

    def __init__(self, env_num, obs_dim=17, act_dim=6, max_episode_steps=1000, seed=0):
        self.E, self.D, self.A, self.max_steps = env_num, obs_dim, act_dim, max_episode_steps
        self.rng = np.random.default_rng(seed)
        self.W = self.rng.standard_normal((act_dim, obs_dim)) * 0.1
        self.x = np.zeros((env_num, obs_dim))
        self.steps = np.zeros(env_num, np.int64)

    def reset_where(self, mask):
        n = int(mask.sum())
        self.x[mask] = self.rng.standard_normal((n, self.D)) * 0.1
        self.steps[mask] = 0
        return self.x

    def reset(self):
        return self.reset_where(np.ones(self.E, bool)).copy()

    def step(self, action):
        a = np.clip(np.asarray(action, np.float64), -1, 1)
        self.x = 0.98 * self.x + a @ self.W + self.rng.standard_normal(self.x.shape) * 0.01
        self.steps += 1
        reward = self.x[:, 0] - 0.1 * (a * a).sum(1)
        done = self.steps >= self.max_steps
        return self.x.copy(), reward, done


class ParallelEnv(object):
    def __init__(self, config, device=None):
        self.config = config
        self.env_num = config['env_num']
        self.device = torch.device('cuda') if device is None else device
        self.continuous_action = config['continuous_action']
        seed = config['seed'] or 0
        if self.continuous_action:
            self.sim = SyntheticHostSim(self.env_num, seed=seed)
            self.norm = DeviceVecNormalize(self.env_num, self.sim.D, gamma=GAMMA, device=self.device)
            self.obs_space, self.act_space = Space((self.sim.D, ), None), Space((self.sim.A, ), None)
            self._max_episode_steps = self.sim.max_steps
        else:
            self.env = DeviceVectorEnv(config['env'], self.env_num, dim=84, horizon=config['step_nums'], seed=seed,
                                       device=self.device)
            self.obs_space, self.act_space = Space((4, 84, 84), None), Space((), self.env.act_dim)
        self.acc = torch.zeros(3, dtype=torch.float64, device=self.device)  # episodes, returns, lengths
        self.ep_ret = np.zeros(self.env_num)

    def reset(self):
        if self.continuous_action:
            return self.norm.reset(self.sim.reset())
        return self.env.reset()

    def step(self, action):
        """-> (next_obs, reward f32 [E], done f32 [E]) device tensors; finished envs are reset and
        return their reset observation (env_utils.py:95-103)"""
        if self.continuous_action:
            raw, rew, done = self.sim.step(action.detach().cpu().numpy())
            self.ep_ret += rew
            obs, r = self.norm.step(raw, rew, done)
            if done.any():
                for k in np.nonzero(done)[0]:
                    self.acc += torch.tensor([1.0, self.ep_ret[k], float(self.sim.steps[k])], dtype=torch.float64,
                                             device=self.device)
                    self.ep_ret[k] = 0
                obs = self.norm.reset_where(done, self.sim.reset_where(done), obs)
            return obs, r, torch.from_numpy(done.astype(np.float32)).to(self.device)
        obs, rew, done, _ = self.env.step(action.long().reshape(-1))
        self.env.accumulate_episode_stats(self.acc)
        return obs, rew, done.float()

    def pop_episode_stats(self):
        n, r, l = [float(x) for x in self.acc.cpu()]
        self.acc.zero_()
        return n, (r / n if n else 0.0), (l / n if n else 0.0)
