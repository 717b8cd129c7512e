"""DeviceVecNormalize — E VecNormalizeEnv instances (parl/env/mujoco_wrappers.py:95-169, one per
environment as examples/PPO/env_utils.py:118-127 builds them) as arrays in HBM.

The environments themselves (MuJoCo, BASELINE configs[4]) step on the host; their raw float64
observations / rewards / dones are uploaded once per step and everything after that — the running
mean / variance / count per env (float64, numpy's operation order: bit-identical to the
reference), clipping, the float32 rows the RolloutStorage keeps — happens in two kernel launches
(`parlhip_vecnorm_obs_f64`, `parlhip_vecnorm_reward_f64`)."""
import numpy as np
import torch

from .. import ops

__all__ = ['DeviceVecNormalize']


class DeviceVecNormalize(object):
    def __init__(self, env_num, obs_dim, ob=True, ret=True, clipob=10., cliprew=10., gamma=0.99, epsilon=1e-8,
                 device=None):
        dev = torch.device('cuda') if device is None else torch.device(device)
        self.device, self.E, self.D = dev, env_num, obs_dim
        self.ob, self.ret_norm = ob, ret
        self.clipob, self.cliprew, self.gamma, self.epsilon = clipob, cliprew, gamma, epsilon
        f64 = lambda fill, *s: torch.full(s, fill, dtype=torch.float64, device=dev)
        # RunningMeanStd(epsilon=1e-4) (mujoco_wrappers.py:78-81)
        self.ob_mean, self.ob_var, self.ob_count = f64(0.0, env_num, obs_dim), f64(1.0, env_num, obs_dim), f64(1e-4, env_num)
        self.ret_mean, self.ret_var, self.ret_count = f64(0.0, env_num), f64(1.0, env_num), f64(1e-4, env_num)
        self.ret = f64(0.0, env_num)
        self.training = True

    def _up(self, x, dtype):
        if not torch.is_tensor(x):
            x = torch.from_numpy(np.ascontiguousarray(np.asarray(x)))
        return x.to(device=self.device, dtype=dtype, non_blocking=True)

    def _obfilt(self, raw, mask=None, out=None):
        raw = self._up(raw, torch.float64).reshape(self.E, self.D)
        if not self.ob:
            o = raw.float()
            return o if out is None else out.copy_(o)
        return ops.vecnorm_obs(raw, self.ob_mean, self.ob_var, self.ob_count, mask=mask, out=out, clipob=self.clipob,
                               eps=self.epsilon, update=self.training)

    def reset(self, raw_obs):
        """VecNormalizeEnv.reset for every env (mujoco_wrappers.py:136-139): ret = 0, filter obs"""
        self.ret.zero_()
        return self._obfilt(raw_obs)

    def step(self, raw_obs, raw_rew, done, out_obs=None):
        """VecNormalizeEnv.step after the host envs stepped (mujoco_wrappers.py:120-134).
        Returns (obs f32 [E,D], rew f32 [E]) device tensors."""
        obs = self._obfilt(raw_obs, out=out_obs)
        rew = self._up(raw_rew, torch.float64).reshape(self.E)
        done = self._up(done, torch.uint8).reshape(self.E)
        if self.ret_norm:
            rew_out = ops.vecnorm_reward(rew, done, self.ret, self.ret_mean, self.ret_var, self.ret_count,
                                         gamma=self.gamma, cliprew=self.cliprew, eps=self.epsilon)
        else:
            self.ret.mul_(self.gamma).add_(rew).masked_fill_(done.bool(), 0.0)
            rew_out = rew.float()
        return obs, rew_out

    def reset_where(self, done, raw_reset_obs, obs):
        """the ParallelEnv auto-reset (examples/PPO/env_utils.py:95-103): envs with done filter
        their reset observation (updating their statistics once more); `obs` rows of those envs are
        overwritten in place."""
        done = self._up(done, torch.uint8).reshape(self.E)
        raw = self._up(raw_reset_obs, torch.float64).reshape(self.E, self.D)
        if not self.ob:
            obs[done.bool()] = raw.float()[done.bool()]
            return obs
        return ops.vecnorm_obs(raw, self.ob_mean, self.ob_var, self.ob_count, mask=done, out=obs, clipob=self.clipob,
                               eps=self.epsilon, update=self.training)

    def get_ob_rms(self):
        return {'mean': self.ob_mean.cpu().numpy(), 'var': self.ob_var.cpu().numpy(), 'count': self.ob_count.cpu().numpy()}

    def set_ob_rms(self, rms):
        self.ob_mean.copy_(self._up(rms['mean'], torch.float64))
        self.ob_var.copy_(self._up(rms['var'], torch.float64))
        self.ob_count.copy_(self._up(rms['count'], torch.float64))

    def train(self):
        self.training = True

    def eval(self):
        self.training = False
