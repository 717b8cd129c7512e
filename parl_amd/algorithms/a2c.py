"""A2C — parl/algorithms/torch/a2c.py:26-90 (ctor `A2C(model, config)`) and
parl/algorithms/paddle/a2c.py:25-118 (ctor `A2C(model, vf_loss_coeff)`); both call styles are
accepted because examples/A2C uses the Paddle one and benchmark/torch/a2c the torch one."""
import torch
import torch.nn.functional as F

from ..core import Algorithm

__all__ = ['A2C']


class A2C(Algorithm):
    def __init__(self, model, config=None, vf_loss_coeff=None):
        if isinstance(config, (int, float)) and vf_loss_coeff is None:  # A2C(model, 0.5)
            vf_loss_coeff, config = config, None
        if config is not None:
            assert isinstance(config['vf_loss_coeff'], (int, float))
            vf_loss_coeff = config['vf_loss_coeff']
            lr = config.get('learning_rate', 0.001)
        else:
            assert isinstance(vf_loss_coeff, (int, float))
            lr = 0.001
        for m in ('value', 'policy', 'policy_and_value'):  # check_model_method, utils.py:217-243
            assert callable(getattr(model, m, None)), '%s: model needs a `%s` method' % (self.__class__.__name__, m)
        super(A2C, self).__init__(model)
        self.vf_loss_coeff = vf_loss_coeff
        self.optimizer = torch.optim.Adam(self.model.parameters(), lr=lr)
        self.config = config
        self.grad_hook = None

    def learn(self, obs, actions, advantages, target_values, lr, entropy_coeff):
        """torch a2c.py:40-81 — sums, clip_grad_norm_(40), returns the four loss scalars."""
        logits, values = self.model.policy_and_value(obs)
        logp_all = F.log_softmax(logits, dim=1)
        actions_log_probs = logp_all.gather(1, actions.unsqueeze(1)).squeeze(1)
        pi_loss = -1.0 * torch.sum(actions_log_probs * advantages)
        delta = values - target_values
        vf_loss = 0.5 * torch.sum(torch.square(delta))
        entropy = torch.sum(-(logp_all.exp() * logp_all).sum(-1))
        total_loss = pi_loss + vf_loss * self.vf_loss_coeff + entropy * entropy_coeff
        for g in self.optimizer.param_groups:
            g['lr'] = lr
        self._zero_grad()
        total_loss.backward()
        if self.grad_hook is not None:
            self.grad_hook(self.model)
        torch.nn.utils.clip_grad_norm_(self.model.parameters(), max_norm=40.0)
        self.optimizer.step()
        # values, not graph nodes: the reference's agents call `.cpu().numpy()` on them (examples/A2C/atari_agent.py:108)
        return total_loss.detach(), pi_loss.detach(), vf_loss.detach(), entropy.detach()

    @torch.no_grad()
    def sample(self, obs):
        logits, values = self.model.policy_and_value(obs)
        return torch.distributions.Categorical(logits=logits).sample().long(), values

    @torch.no_grad()
    def prob_and_value(self, obs):
        logits, values = self.model.policy_and_value(obs)
        return F.softmax(logits, dim=1), values

    @torch.no_grad()
    def predict(self, obs):
        return self.model.policy(obs).max(-1)[1]

    @torch.no_grad()
    def value(self, obs):
        return self.model.value(obs)
