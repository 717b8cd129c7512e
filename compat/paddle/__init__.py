"""`import paddle` for the reference's OWN example scripts (examples/IMPALA, examples/A2C), served by PyTorch-ROCm.

The reference's IMPALA / A2C examples are written against Paddle: `paddle.to_tensor`, `paddle.squeeze`,
`paddle.nn.{Conv2D, Linear, Flatten}`, `paddle.nn.functional.relu`, `paddle.ParamAttr(initializer=
paddle.nn.initializer.Normal())`, `paddle.io.DataLoader.from_generator` — that is ALL of Paddle their five files
touch (examples/IMPALA/{train,actor,atari_model,atari_agent,impala_config}.py; the algorithms behind
`parl.algorithms.{IMPALA, A2C}` are this repository's).  This package maps exactly that surface onto torch on
the default device, so that — together with compat/parl and compat/gym — those files run UNMODIFIED on the
MI355X path (tests/test_reference_scripts.py).  It is an import alias for the host framework the north star
names, not a second backend: there is no Paddle here and nothing falls back to it.

Semantics kept: `to_tensor(x, dtype=...)` returns a tensor on the default device (Paddle places tensors on the
GPU when there is one); layers create their parameters there too, with PADDLE's default initialisation where the
script names none (Conv2D: Normal(0, sqrt(2 / (k_h k_w in_channels))), Linear: Xavier uniform, biases 0 — torch's own
defaults are 2.3-2.4x smaller per layer, which matters for how the A2C example starts); `Normal()` is N(0, 1)
(paddle.nn.initializer.Normal defaults); `Conv2D` is this repository's GEMM-lowered convolution (the image ships
no MIOpen kernel database for gfx950).  Parameter LAYOUT differs from Paddle's for `Linear` (torch keeps
[out, in]) — invisible to the examples, which exchange weights only between their own models.

`paddle.nn`, `paddle.nn.functional`, `paddle.nn.initializer` and `paddle.io` are module objects built here
(registered in sys.modules so that `import paddle.nn.functional as F` works)."""
import sys
import types

import numpy as np
import torch

from parl_amd.models.atari_model import GemmConv2d

__version__ = '2.3.1'  # the reference CI's Paddle (.teamcity/build.sh:208)

_DTYPES = {'float32': torch.float32, 'float64': torch.float64, 'int64': torch.int64, 'int32': torch.int32,
           'bool': torch.bool, 'uint8': torch.uint8, 'float16': torch.float16}


def _device():
    return torch.device('cuda') if torch.cuda.is_available() else torch.device('cpu')


def _dtype(dtype):
    if dtype is None or isinstance(dtype, torch.dtype):
        return dtype
    return _DTYPES[str(dtype).replace('paddle.', '')]


def to_tensor(data, dtype=None, place=None, stop_gradient=True):
    t = data if isinstance(data, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(data))
    return t.to(device=_device(), dtype=_dtype(dtype))


def squeeze(x, axis=None):
    return x.squeeze() if axis is None else x.squeeze(axis)


class ParamAttr(object):
    def __init__(self, name=None, initializer=None, learning_rate=1.0, regularizer=None, trainable=True):
        self.initializer = initializer


no_grad = torch.no_grad
Tensor = torch.Tensor


# ---- paddle.nn.initializer ----
class Normal(object):
    """paddle.nn.initializer.Normal(mean=0.0, std=1.0)"""

    def __init__(self, mean=0.0, std=1.0):
        self.mean, self.std = mean, std

    def __call__(self, tensor):
        torch.nn.init.normal_(tensor, self.mean, self.std)


def _apply(attr, tensor, default=None):
    """the attribute's initializer, else Paddle's DEFAULT for that parameter (torch's own defaults are 2.3-2.4x smaller
    per layer; see parl_amd.models.atari_model.paddle_default_init_ for what that does to the A2C example's start)"""
    init = getattr(attr, 'initializer', None) if attr is not None else None
    init = init if init is not None else default
    if init is not None:
        with torch.no_grad():
            init(tensor)


# ---- paddle.nn ----
class Conv2D(GemmConv2d):
    """paddle.nn.Conv2D(in_channels, out_channels, kernel_size, stride=1, padding=0, ...)"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, weight_attr=None, bias_attr=None):
        super(Conv2D, self).__init__(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                                     device=_device())
        # Paddle's defaults: weight ~ Normal(0, sqrt(2 / (k_h k_w in_channels))), bias 0
        fan_in = self.in_channels * self.kernel_size[0] * self.kernel_size[1]
        _apply(weight_attr, self.weight, lambda t: t.normal_(0.0, (2.0 / fan_in) ** 0.5))
        _apply(bias_attr, self.bias, lambda t: t.zero_())


class Linear(torch.nn.Linear):
    """paddle.nn.Linear(in_features, out_features, weight_attr=None, bias_attr=None)"""

    def __init__(self, in_features, out_features, weight_attr=None, bias_attr=None, name=None):
        super(Linear, self).__init__(in_features, out_features, device=_device())
        # Paddle's defaults: weight ~ Xavier uniform, bias 0
        _apply(weight_attr, self.weight, torch.nn.init.xavier_uniform_)
        _apply(bias_attr, self.bias, lambda t: t.zero_())


# ---- paddle.io ----
class _GeneratorLoader(object):
    """paddle.io.DataLoader.from_generator as examples/IMPALA/train.py:129-133 uses it: a bounded prefetch queue in
    front of a batch generator.  Here the generator is simply iterated (its batches are numpy arrays that
    agent.learn uploads itself, atari_agent.py:58-63); the learner thread blocks in the generator's own queue."""

    def __init__(self, capacity):
        self.capacity = capacity
        self._reader = None

    def set_batch_generator(self, reader, places=None):
        self._reader = reader
        return self

    def __call__(self):
        return iter(self._reader())

    __iter__ = __call__


class DataLoader(object):
    @staticmethod
    def from_generator(feed_list=None, capacity=None, use_double_buffer=True, iterable=True, return_list=False,
                       use_multiprocess=False, drop_last=True):
        return _GeneratorLoader(capacity)


def _module(name, **members):
    m = types.ModuleType(name)
    m.__dict__.update(members)
    sys.modules[name] = m
    return m


_functional = _module('paddle.nn.functional', relu=torch.nn.functional.relu, softmax=torch.nn.functional.softmax,
                      log_softmax=torch.nn.functional.log_softmax)
_initializer = _module('paddle.nn.initializer', Normal=Normal)
nn = _module('paddle.nn', Layer=torch.nn.Module, Flatten=torch.nn.Flatten, Conv2D=Conv2D, Linear=Linear,
             functional=_functional, initializer=_initializer)
io = _module('paddle.io', DataLoader=DataLoader)
