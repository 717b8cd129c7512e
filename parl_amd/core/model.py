"""parl.Model (torch flavour) — same contract as parl/core/torch/model.py:24-134 and
parl/core/model_base.py:16-55: an nn.Module with get_weights() -> {name: np.ndarray},
set_weights(dict) and sync_weights_to(target, decay)."""
import numpy as np
import torch
import torch.nn as nn

__all__ = ['Model']


class Model(nn.Module):
    def sync_weights_to(self, target_model, decay=0.0):
        """target = decay * target + (1 - decay) * self   (core/torch/model.py:76-113)"""
        assert target_model is not self, 'cannot copy between identical model'
        assert isinstance(target_model, Model)
        assert self.__class__.__name__ == target_model.__class__.__name__, \
            'must be the same class for params syncing!'
        assert 0 <= decay <= 1
        target_vars = dict(target_model.named_parameters())
        with torch.no_grad():
            for name, var in self.named_parameters():
                target_vars[name].data.copy_(decay * target_vars[name].data + (1 - decay) * var.data)
        _layouts_follow(target_model)   # (.data writes do not move the autograd version counters)

    def get_weights(self):
        """{name: host numpy copy} (core/torch/model.py:115-123)"""
        # always a COPY: on a CPU-resident model `.numpy()` would alias the live parameters
        return {k: (v.detach().cpu().numpy() if v.is_cuda else v.detach().numpy().copy())
                for k, v in self.state_dict().items()}

    def set_weights(self, weights):
        """load host numpy weights (core/torch/model.py:125-134)"""
        self.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in weights.items()})
        _layouts_follow(self)


def _layouts_follow(model):
    """a model that keeps derived device copies of its weights (AtariModel42's MFMA operand-order buffer) rebuilds
    them after a bulk write"""
    f = getattr(model, 'refresh_actor_layout', None)
    if f is not None:
        f()
