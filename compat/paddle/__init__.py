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
GPU when there is one); layers create their parameters there too; `Normal()` is N(0, 1)
(paddle.nn.initializer.Normal defaults); `Conv2D` is this repository's GEMM-lowered convolution (the image ships
no MIOpen kernel database for gfx950).  Parameter LAYOUT differs from Paddle's for `Linear` (torch keeps
[out, in]) — invisible to the examples, which exchange weights only between their own models."""
import numpy as np
import torch

from . import io, nn  # noqa: F401

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
    if isinstance(data, torch.Tensor):
        t = data
    else:
        t = torch.from_numpy(np.ascontiguousarray(data))
    return t.to(device=_device(), dtype=_dtype(dtype))


def squeeze(x, axis=None):
    return x.squeeze() if axis is None else x.squeeze(axis)


class ParamAttr(object):
    def __init__(self, name=None, initializer=None, learning_rate=1.0, regularizer=None, trainable=True):
        self.initializer = initializer


no_grad = torch.no_grad
Tensor = torch.Tensor
