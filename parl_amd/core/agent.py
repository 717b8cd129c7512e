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

    def attach(self, **stateful):
        """Register objects with state_dict() / load_state_dict() (DeviceVectorEnv, DeviceRollout,
        DeviceA2CRollout, ...) whose state is checkpointed next to the model: with actors resident
        on the GPU the envs and the sampling RNG live in this process, and a model-only
        checkpoint (all the reference needs, its actors being separate processes) could not
        resume a run.  SURVEY 8 f4."""
        if not hasattr(self, '_stateful'):
            self._stateful = {}
        for k, v in stateful.items():
            assert hasattr(v, 'state_dict') and hasattr(v, 'load_state_dict'), k
            self._stateful[k] = v

    def save(self, save_path, model=None):
        """torch.save(model.state_dict()) exactly as core/torch/agent.py:100-124 (the file is
        interchangeable with the reference's); attached env / sampler state goes to
        `save_path + '.env'`."""
        if model is None:
            model = self.alg.model
        dirname = os.sep.join(save_path.split(os.sep)[:-1])
        if dirname != '' and not os.path.exists(dirname):
            os.makedirs(dirname)
        extra = getattr(self, '_stateful', None)
        # the attached objects' state first: their state_dict() drains the streams that write it (and the
        # learner stream that writes the parameters), so the model saved next is not torn either
        blob = {k: v.state_dict() for k, v in extra.items()} if extra else None
        torch.save(model.state_dict(), save_path)
        if extra:
            torch.save(blob, save_path + '.env')
        elif os.path.exists(save_path + '.env'):
            os.remove(save_path + '.env')  # an older run's env state must not pair up with these weights

    def restore(self, save_path, model=None, map_location=None):
        if model is None:
            model = self.alg.model
        model.load_state_dict(torch.load(save_path, map_location=map_location))
        extra = getattr(self, '_stateful', None)
        if extra:
            if not os.path.exists(save_path + '.env'):
                raise FileNotFoundError('%s.env: env / sampler state is attached to this agent but the checkpoint '
                                        'holds none (a model-only checkpoint would silently restart the envs)' % save_path)
            # tensors, numbers, strings, lists, dicts only: no arbitrary unpickling
            blob = torch.load(save_path + '.env', map_location='cpu', weights_only=True)
            for k, v in extra.items():
                if k not in blob:
                    raise KeyError('%s.env holds no state for %r' % (save_path, k))
                v.load_state_dict(blob[k])

    def train(self):
        self.alg.model.train()
        self.training = True

    def eval(self):
        self.alg.model.eval()
        self.training = False
