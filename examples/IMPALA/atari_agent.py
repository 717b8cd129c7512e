"""torch twin of examples/IMPALA/atari_agent.py:21-100.  Same methods; tensors stay on the GPU
(numpy inputs are accepted and uploaded, as in the reference)."""
import numpy as np
import torch

import parl_amd as parl
from parl_amd import ops


def _dev(x, dtype, device):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    return x.to(device=device, dtype=dtype)


class AtariAgent(parl.Agent):
    def __init__(self, algorithm, seed=0, device=None):
        super(AtariAgent, self).__init__(algorithm)
        self.device = torch.device(device if device is not None else 'cuda')
        self.alg.model.to(self.device)
        self.seed = seed
        self._sample_calls = 0

    def sample(self, obs):
        """obs [B,4,H,W] (uint8 or float32) -> (actions int64 [B], behaviour_logits f32 [B,A]).
        Reference: softmax on the device, np.random.choice per env on the host
        (atari_agent.py:35-42); here both happen in one HIP kernel (same arithmetic)."""
        obs = _dev(obs, None, self.device) if not isinstance(obs, torch.Tensor) else obs.to(self.device)
        probs, logits = self.alg.sample(obs)
        actions = ops.policy_sample(logits, self.seed, self._sample_calls)
        self._sample_calls += 1
        return actions, logits

    def learn(self, obs, actions, behaviour_logits, rewards, dones, lr, entropy_coeff, time_major=False):
        d = self.device
        vtrace_loss, kl = self.alg.learn(
            _dev(obs, None, d) if not isinstance(obs, torch.Tensor) else obs, _dev(actions, torch.int64, d),
            _dev(behaviour_logits, torch.float32, d), _dev(rewards, torch.float32, d), _dev(dones, torch.bool, d), lr,
            entropy_coeff, time_major=time_major)
        out = torch.stack([vtrace_loss.total_loss, vtrace_loss.pi_loss, vtrace_loss.vf_loss, vtrace_loss.entropy,
                           kl]).detach().cpu().numpy()  # ONE device->host copy per update
        return tuple(out)
