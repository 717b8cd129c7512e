"""The A2C example's network starts the way the reference's does: examples/A2C/atari_model.py:21-104 names no
initializer, so its layers take Paddle's defaults (nn.Conv2D: Normal(0, sqrt(2 / (k_h k_w in_channels))), nn.Linear:
Xavier uniform, biases 0) — not torch's kaiming_uniform(a=sqrt(5)), under which the first Adam steps at the reference's
learning rate could switch the whole 512-unit layer off (profiles/README.md, the r04 A2C rows).  CPU."""
import math

import torch


def test_atari_model84_takes_paddles_default_initialisation():
    from parl_amd.models import AtariModel84
    torch.manual_seed(0)
    m = AtariModel84(6)
    for conv in (m.conv1, m.conv2, m.conv3):
        fan_in = conv.in_channels * conv.kernel_size[0] * conv.kernel_size[1]
        std = math.sqrt(2.0 / fan_in)
        assert abs(float(conv.weight.detach().std()) - std) < 0.05 * std
        assert abs(float(conv.weight.detach().mean())) < 0.05 * std
        assert float(conv.weight.detach().abs().max()) > 2.5 * std          # a normal, not a bounded uniform
        assert float(conv.bias.detach().abs().max()) == 0.0
    for fc in (m.fc, m.policy_fc, m.value_fc):
        bound = math.sqrt(6.0 / (fc.in_features + fc.out_features))
        assert float(fc.weight.detach().abs().max()) <= bound * (1 + 1e-6)
        assert float(fc.weight.detach().abs().max()) > 0.9 * bound or fc.weight.numel() < 1000
        assert float(fc.bias.detach().abs().max()) == 0.0
    # the hidden layer is alive on a picture: torch's defaults left about two thirds of its units at zero on every input
    x = torch.zeros((4, 4, 84, 84), dtype=torch.float32)
    x[:, :, 20:60, 30:50] = 200.0
    with torch.no_grad():
        h = m._trunk(x)
    assert float((h > 0).float().mean()) > 0.3


def test_compat_paddle_layers_take_paddles_defaults_unless_told_otherwise():
    import importlib
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'compat'))
    try:
        paddle = importlib.import_module('paddle')
        torch.manual_seed(0)
        c = paddle.nn.Conv2D(in_channels=4, out_channels=32, kernel_size=8, stride=4, padding=1)
        std = math.sqrt(2.0 / (4 * 64))
        assert abs(float(c.weight.detach().std()) - std) < 0.05 * std and float(c.bias.detach().abs().max()) == 0.0
        fc = paddle.nn.Linear(in_features=256, out_features=6)
        assert float(fc.weight.detach().abs().max()) <= math.sqrt(6.0 / 262) * (1 + 1e-6) and float(fc.bias.detach().abs().max()) == 0.0
        n = paddle.nn.Linear(in_features=256, out_features=6,
                             weight_attr=paddle.ParamAttr(initializer=paddle.nn.initializer.Normal()),
                             bias_attr=paddle.ParamAttr(initializer=paddle.nn.initializer.Normal()))
        assert 0.8 < float(n.weight.detach().std()) < 1.2 and float(n.bias.detach().abs().max()) > 0.0   # N(0, 1) as asked for
    finally:
        sys.path.remove(os.path.join(root, 'compat'))
        for k in [k for k in sys.modules if k == 'paddle' or k.startswith('paddle.')]:
            del sys.modules[k]
