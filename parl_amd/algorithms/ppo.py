"""PPO — parl/algorithms/torch/ppo.py:27-206 (same constructor, same learn / sample / predict /
value contract; the Paddle twin parl/algorithms/paddle/ppo.py:28-216 has identical arithmetic).

What runs where: the model forward / backward, Normal / Categorical log-probabilities, the clipped
surrogate and Adam are PyTorch-ROCm; the per-minibatch advantage normalisation
`(adv - adv.mean()) / (adv.std() + 1e-8)` (ppo.py:124-127, unbiased std) is the HIP kernel
`parlhip_adv_normalize_f32` (advantages carry no gradient).  The GAE scan and the minibatch
gather that feed learn() live in parl_amd.storage.RolloutStorage."""
import torch
import torch.nn as nn
from torch.distributions import Categorical, Normal

from .. import ops
from ..core import Algorithm

__all__ = ['PPO']


class PPO(Algorithm):
    def __init__(self,
                 model,
                 clip_param=0.1,
                 value_loss_coef=0.5,
                 entropy_coef=0.01,
                 initial_lr=2.5e-4,
                 eps=1e-5,
                 max_grad_norm=0.5,
                 use_clipped_value_loss=True,
                 norm_adv=True,
                 continuous_action=False):
        for m in ('value', 'policy'):  # check_model_method, parl/utils/utils.py:217-243
            assert callable(getattr(model, m, None)), '%s: model needs a `%s` method' % (self.__class__.__name__, m)
        # the argument checks of ppo.py:57-65
        assert isinstance(clip_param, float)
        assert isinstance(value_loss_coef, float)
        assert isinstance(entropy_coef, float)
        assert isinstance(initial_lr, float)
        assert isinstance(eps, float)
        assert isinstance(max_grad_norm, float)
        assert isinstance(use_clipped_value_loss, bool)
        assert isinstance(norm_adv, bool)
        assert isinstance(continuous_action, bool)
        super(PPO, self).__init__(model)
        self.clip_param = clip_param
        self.value_loss_coef = value_loss_coef
        self.entropy_coef = entropy_coef
        self.max_grad_norm = max_grad_norm
        self.use_clipped_value_loss = use_clipped_value_loss
        self.norm_adv = norm_adv
        self.continuous_action = continuous_action
        if torch.cuda.is_available():  # ppo.py:75-76
            self.model = model.to(torch.device('cuda'))
        self.optimizer = torch.optim.Adam(self.model.parameters(), lr=initial_lr, eps=eps)
        self.grad_hook = None  # parl_amd.dist.FlatGradAllReduce(model, average=True) for data-parallel learners

    def _dist(self, obs):
        if self.continuous_action:
            mean, std = self.model.policy(obs)
            return Normal(mean, std)
        return Categorical(logits=self.model.policy(obs))

    def _losses(self, obs, action, old_value, ret, old_logp, adv):
        """the three terms of the PPO objective for one minibatch (ppo.py:104-148)"""
        dist = self._dist(obs)
        logp, entropy = dist.log_prob(action), dist.entropy()
        if self.continuous_action:  # independent Normal per action dimension
            logp, entropy = logp.sum(1), entropy.sum(1)
        if self.norm_adv:  # ppo.py:124-127 on the device kernel (unbiased std, + 1e-8); no gradient
            adv = ops.adv_normalize(adv.detach(), eps=1e-8).view_as(adv)
        eps = self.clip_param
        ratio = torch.exp(logp - old_logp)
        policy_loss = -torch.minimum(ratio * adv, ratio.clamp(1.0 - eps, 1.0 + eps) * adv).mean()
        value = self.model.value(obs).view(-1)
        if self.use_clipped_value_loss:
            clipped = old_value + (value - old_value).clamp(-eps, eps)
            value_loss = 0.5 * torch.maximum((value - ret)**2, (clipped - ret)**2).mean()
        else:
            value_loss = 0.5 * ((ret - value)**2).mean()
        return value_loss, policy_loss, entropy.mean()

    def learn(self, batch_obs, batch_action, batch_value, batch_return, batch_logprob, batch_adv, lr=None):
        """ppo.py:81-158: one clipped-surrogate update on a minibatch.  Returns
        (value_loss, action_loss, entropy_loss) as python floats."""
        value_loss, action_loss, entropy_loss = self._losses(batch_obs, batch_action, batch_value, batch_return,
                                                             batch_logprob, batch_adv)
        loss = value_loss * self.value_loss_coef + action_loss - entropy_loss * self.entropy_coef
        if lr:
            for group in self.optimizer.param_groups:
                group['lr'] = lr
        self._zero_grad()
        loss.backward()
        if self.grad_hook is not None:
            self.grad_hook(self.model)
        nn.utils.clip_grad_norm_(self.model.parameters(), self.max_grad_norm)
        self.optimizer.step()
        return value_loss.item(), action_loss.item(), entropy_loss.item()

    @torch.no_grad()
    def sample(self, obs):
        """ppo.py:160-187: (value, action, action_log_probs, action_entropy)"""
        value = self.model.value(obs)
        dist = self._dist(obs)
        action = dist.sample()
        if self.continuous_action:
            return value, action, dist.log_prob(action).sum(1), dist.entropy().sum(1)
        return value, action, dist.log_prob(action), dist.entropy()

    @torch.no_grad()
    def predict(self, obs):
        """ppo.py:189-203: the mean action / the argmax of the probabilities (keepdim)"""
        if self.continuous_action:
            action, _ = self.model.policy(obs)
            return action
        return Categorical(logits=self.model.policy(obs)).probs.argmax(dim=-1, keepdim=True)

    @torch.no_grad()
    def value(self, obs):
        return self.model.value(obs)
