"""AtariModel — examples/PPO/atari_model.py:21-68 (conv 32k8s4 / 64k4s2 / 64k3s1, fc 512, no padding)
as a torch parl.Model on GEMM-lowered convolutions (parl_amd.models.GemmConv2d)."""
import torch.nn as nn
import torch.nn.functional as F

import parl_amd as parl
from parl_amd.models import GemmConv2d


class AtariModel(parl.Model):
    def __init__(self, obs_space, act_space):
        super(AtariModel, self).__init__()
        self.conv1 = GemmConv2d(4, 32, 8, stride=4)
        self.conv2 = GemmConv2d(32, 64, 4, stride=2)
        self.conv3 = GemmConv2d(64, 64, 3, stride=1)
        self.fc = nn.Linear(64 * 7 * 7, 512)
        self.fc_pi = nn.Linear(512, act_space.n)
        self.fc_v = nn.Linear(512, 1)

    def _body(self, obs):
        out = F.relu(self.conv1(obs.float() / 255.0))
        out = F.relu(self.conv2(out))
        out = F.relu(self.conv3(out))
        return F.relu(self.fc(out.flatten(1)))

    def value(self, obs):
        return self.fc_v(self._body(obs))

    def policy(self, obs):
        return self.fc_pi(self._body(obs))
