"""PPOAgent — examples/PPO/agent.py:21-121 on device tensors: same learn() loop (update_epochs x
shuffled minibatches of batch_size // num_minibatches, agent.py:91-110), the minibatch gather is
one kernel launch (RolloutStorage.sample_batch), nothing is copied to the host per minibatch."""
import numpy as np
import torch

import parl_amd as parl
from parl_amd.utils import LinearDecayScheduler


class PPOAgent(parl.Agent):
    def __init__(self, algorithm, config):
        super(PPOAgent, self).__init__(algorithm)
        self.config = config
        self.device = next(algorithm.model.parameters()).device
        if self.config['lr_decay']:
            self.lr_scheduler = LinearDecayScheduler(self.config['initial_lr'], self.config['num_updates'])
        self.rng = np.random.RandomState(config.get('seed') or 0)

    def _t(self, obs):
        if not torch.is_tensor(obs):
            obs = torch.from_numpy(np.ascontiguousarray(obs))
        return obs.to(self.device)

    def predict(self, obs):
        return self.alg.predict(self._t(obs).float().unsqueeze(0))[0]

    def sample(self, obs):
        """(value [E,1], action, log-prob [E], entropy [E]) as device tensors (agent.py:46-60 returns numpy)"""
        obs = self._t(obs)
        return self.alg.sample(obs if obs.dtype == torch.uint8 else obs.float())

    def value(self, obs):
        obs = self._t(obs)
        return self.alg.value(obs if obs.dtype == torch.uint8 else obs.float())

    def learn(self, rollout):
        value_loss_epoch = action_loss_epoch = entropy_loss_epoch = 0
        lr = self.lr_scheduler.step(step_num=1) if self.config['lr_decay'] else None
        minibatch_size = int(self.config['batch_size'] // self.config['num_minibatches'])
        indexes = np.arange(self.config['batch_size'])
        for epoch in range(self.config['update_epochs']):
            self.rng.shuffle(indexes)
            for start in range(0, self.config['batch_size'], minibatch_size):
                sample_idx = indexes[start:start + minibatch_size]
                batch_obs, batch_action, batch_logprob, batch_adv, batch_return, batch_value = \
                    rollout.sample_batch(sample_idx)
                if not self.config['continuous_action']:
                    batch_action = batch_action.long()
                value_loss, action_loss, entropy_loss = self.alg.learn(batch_obs, batch_action, batch_value,
                                                                       batch_return, batch_logprob, batch_adv, lr)
                value_loss_epoch += value_loss
                action_loss_epoch += action_loss
                entropy_loss_epoch += entropy_loss
        update_steps = self.config['update_epochs'] * self.config['batch_size']
        return value_loss_epoch / update_steps, action_loss_epoch / update_steps, entropy_loss_epoch / update_steps, lr
