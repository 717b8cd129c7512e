"""RolloutStorage — examples/PPO/storage.py:18-76 with the arrays resident in HBM.

Same constructor, same `append` ring (cur_step wraps, storage.py:35-43), same
`compute_returns(value, done, gamma, gae_lambda)` and `sample_batch(idx)`; the arithmetic runs in
libparl_hip.so:
  * compute_returns = ONE launch of the GAE scan in the "done starts the step" convention
    (`parlhip_gae_f32`, fp32 operation order of storage.py:48-61 kept: bit-exact with numpy;
    `chunked=True` takes the chunk-parallel plan for long T over few sequences, <= 1e-6 relative),
  * sample_batch = ONE launch gathering the six flattened arrays by the minibatch index
    (`parlhip_ppo_sample_batch_f32`).
append accepts host (numpy) rows — MuJoCo steps on the CPU (BASELINE configs[4]) — or device
tensors (e.g. the float32 rows parl_amd.env.DeviceVecNormalize already wrote on the GPU)."""
import numpy as np
import torch

from . import ops

__all__ = ['RolloutStorage']


class RolloutStorage(object):
    def __init__(self, step_nums, env_num, obs_space, act_space, device=None):
        dev = torch.device('cuda') if device is None else torch.device(device)
        self.device = dev
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
        self.obs = z(step_nums, env_num, *obs_space.shape)
        self.actions = z(step_nums, env_num, *act_space.shape)
        self.logprobs = z(step_nums, env_num)
        self.rewards = z(step_nums, env_num)
        self.dones = z(step_nums, env_num)
        self.values = z(step_nums, env_num)
        self.step_nums = step_nums
        self.env_num = env_num
        self.obs_space = obs_space
        self.act_space = act_space
        self.cur_step = 0
        self.advantages = None
        self.returns = None

    def _row(self, x, like):
        if not torch.is_tensor(x):
            x = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float32)))
        return x.to(device=self.device, dtype=torch.float32, non_blocking=True).reshape(like.shape)

    def append(self, obs, action, logprob, reward, done, value):
        t = self.cur_step
        self.obs[t].copy_(self._row(obs, self.obs[t]))
        self.actions[t].copy_(self._row(action, self.actions[t]))
        self.logprobs[t].copy_(self._row(logprob, self.logprobs[t]))
        self.rewards[t].copy_(self._row(reward, self.rewards[t]))
        self.dones[t].copy_(self._row(done, self.dones[t]))
        self.values[t].copy_(self._row(value, self.values[t]))
        self.cur_step = (self.cur_step + 1) % self.step_nums

    def compute_returns(self, value, done, gamma=0.99, gae_lambda=0.95, chunked=False):
        """storage.py:45-64.  value [E] = V(next_obs), done [E] = done flags of the step after the
        rollout.  Returns (advantages, returns) [T,E] device tensors (also kept on self)."""
        value = self._row(value, self.values[0])
        done = self._row(done, self.dones[0])
        adv, ret = ops.gae(self.rewards, self.values, self.dones, value, gamma, gae_lambda, last_done=done,
                           done_convention=ops.GAE_DONE_STARTS_STEP, allow_chunked=chunked)
        self.advantages, self.returns = adv, ret
        return adv, ret

    def sample_batch(self, idx):
        """storage.py:66-76: (obs, actions, logprobs, advantages, returns, values)[idx] of the
        flattened rollout; idx = numpy / torch int64 indices into [0, T*E)."""
        if not torch.is_tensor(idx):
            idx = torch.from_numpy(np.ascontiguousarray(np.asarray(idx, dtype=np.int64)))
        idx = idx.to(device=self.device, dtype=torch.int64)
        n = self.step_nums * self.env_num
        return ops.ppo_sample_batch(self.obs.reshape((n, ) + tuple(self.obs_space.shape)),
                                    self.actions.reshape((n, ) + tuple(self.act_space.shape)),
                                    self.logprobs.reshape(-1), self.advantages.reshape(-1), self.returns.reshape(-1),
                                    self.values.reshape(-1), idx)
