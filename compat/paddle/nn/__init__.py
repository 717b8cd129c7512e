"""paddle.nn as far as the reference's IMPALA / A2C examples use it (see compat/paddle/__init__.py)."""
import torch

from parl_amd.models.atari_model import GemmConv2d

from . import functional, initializer  # noqa: F401

Layer = torch.nn.Module
Flatten = torch.nn.Flatten


def _device():
    return torch.device('cuda') if torch.cuda.is_available() else torch.device('cpu')


def _apply(attr, tensor):
    init = getattr(attr, 'initializer', None) if attr is not None else None
    if init is not None:
        with torch.no_grad():
            init(tensor)


class Conv2D(GemmConv2d):
    """paddle.nn.Conv2D(in_channels, out_channels, kernel_size, stride=1, padding=0, ...)"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, weight_attr=None, bias_attr=None):
        super(Conv2D, self).__init__(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                                     device=_device())
        _apply(weight_attr, self.weight)
        _apply(bias_attr, self.bias)


class Linear(torch.nn.Linear):
    """paddle.nn.Linear(in_features, out_features, weight_attr=None, bias_attr=None)"""

    def __init__(self, in_features, out_features, weight_attr=None, bias_attr=None, name=None):
        super(Linear, self).__init__(in_features, out_features, device=_device())
        _apply(weight_attr, self.weight)
        _apply(bias_attr, self.bias)
