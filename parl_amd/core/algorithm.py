"""parl.Algorithm (torch flavour) — parl/core/torch/algorithm.py:24-92 and the ModelBase walk of
parl/core/algorithm_base.py:30-96 (models found directly in __dict__ or one level inside a
list / tuple / dict)."""
from .model import Model

__all__ = ['Algorithm']


class Algorithm(object):
    def __init__(self, model=None):
        assert isinstance(model, Model)
        self.model = model

    def get_weights(self):
        out = {}
        for key, value in self.__dict__.items():
            if isinstance(value, Model):
                out[key] = value.get_weights()
            elif isinstance(value, (list, tuple)):
                ws = [x.get_weights() for x in value if isinstance(x, Model)]
                if ws:
                    out[key] = ws
            elif isinstance(value, dict):
                ws = {k: x.get_weights() for k, x in value.items() if isinstance(x, Model)}
                if ws:
                    out[key] = ws
        # the torch Algorithm of the reference returns the bare model dict when the only model is
        # `self.model` (core/torch/algorithm.py:57-66)
        if set(out) == {'model'}:
            return out['model']
        return out

    def set_weights(self, weights):
        models = {k: v for k, v in self.__dict__.items() if isinstance(v, Model)}
        if set(models) == {'model'} and 'model' not in weights:
            self.model.set_weights(weights)
            return
        for key, value in self.__dict__.items():
            if isinstance(value, Model):
                assert key in weights, 'weights is inconsistent with current algorithm.'
                value.set_weights(weights[key])
            elif isinstance(value, (list, tuple)):
                ms = [x for x in value if isinstance(x, Model)]
                if ms:
                    assert key in weights and len(ms) == len(weights[key]), \
                        'weights is inconsistent with current algorithm.'
                    for m, w in zip(ms, weights[key]):
                        m.set_weights(w)
            elif isinstance(value, dict):
                ms = {k: x for k, x in value.items() if isinstance(x, Model)}
                if ms:
                    assert key in weights and set(ms) == set(weights[key]), \
                        'weights is inconsistent with current algorithm.'
                    for k, m in ms.items():
                        m.set_weights(weights[key][k])

    def _zero_grad(self):
        """reset gradients before backward.  With a data-parallel grad_hook (parl_amd.dist.
        FlatGradAllReduce) the gradients are views of its flat bucket: zero the bucket in place
        instead of dropping the views."""
        hook = getattr(self, 'grad_hook', None)
        if hook is not None and hasattr(hook, 'zero_grad'):
            hook.zero_grad()
        else:
            self.optimizer.zero_grad(set_to_none=True)

    def learn(self, *args, **kwargs):
        raise NotImplementedError

    def predict(self, *args, **kwargs):
        raise NotImplementedError

    def sample(self, *args, **kwargs):
        raise NotImplementedError
