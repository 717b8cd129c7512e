"""MujocoModel — examples/PPO/mujoco_model.py:21-63 as a torch parl.Model."""
import numpy as np
import torch
import torch.nn as nn

import parl_amd as parl


class MujocoModel(parl.Model):
    def __init__(self, obs_space, act_space):
        super(MujocoModel, self).__init__()
        self.fc1 = nn.Linear(obs_space.shape[0], 64)
        self.fc2 = nn.Linear(64, 64)
        self.fc_value = nn.Linear(64, 1)
        self.fc_policy = nn.Linear(64, int(np.prod(act_space.shape)))
        self.fc_pi_std = nn.Parameter(torch.zeros(1, int(np.prod(act_space.shape))))  # Constant(0) log-std

    def _body(self, obs):
        return torch.tanh(self.fc2(torch.tanh(self.fc1(obs))))

    def value(self, obs):
        return self.fc_value(self._body(obs))

    def policy(self, obs):
        return self.fc_policy(self._body(obs)), torch.exp(self.fc_pi_std)
