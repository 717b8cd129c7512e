"""parl.Agent (torch flavour) — parl/core/torch/agent.py:29-176, parl/core/agent_base.py:16-89."""
import os

import torch

from .algorithm import Algorithm

__all__ = ['Agent']


class Agent(object):
    def __init__(self, algorithm):
        assert isinstance(algorithm, Algorithm)
        self.alg = algorithm
        self.training = True

    def get_weights(self, *args, **kwargs):
        return self.alg.get_weights(*args, **kwargs)

    def set_weights(self, weights, *args, **kwargs):
        self.alg.set_weights(weights, *args, **kwargs)

    def learn(self, *args, **kwargs):
        raise NotImplementedError

    def predict(self, *args, **kwargs):
        raise NotImplementedError

    def sample(self, *args, **kwargs):
        raise NotImplementedError

    def save(self, save_path, model=None):
        """torch.save(model.state_dict()) (core/torch/agent.py:100-124)"""
        if model is None:
            model = self.alg.model
        dirname = os.sep.join(save_path.split(os.sep)[:-1])
        if dirname != '' and not os.path.exists(dirname):
            os.makedirs(dirname)
        torch.save(model.state_dict(), save_path)

    def restore(self, save_path, model=None, map_location=None):
        if model is None:
            model = self.alg.model
        model.load_state_dict(torch.load(save_path, map_location=map_location))

    def train(self):
        self.alg.model.train()
        self.training = True

    def eval(self):
        self.alg.model.eval()
        self.training = False
