"""parl_amd.algorithms.A2C against the reference's own torch A2C.learn.

tests/golden/a2c_learn.npz was produced by importing parl/algorithms/torch/a2c.py:40-81 and
benchmark/torch/a2c/atari_model.py:23-104 from /root/reference (make_a2c_golden.py).  The CPU test
checks the host logic of A2C.learn (both constructor styles: paddle `A2C(model, vf_loss_coeff)` and
torch `A2C(model, config)`) on a stock-torch twin of the network; the -m gpu test runs the product
model (AtariModel84, HIP convolutions) on the device.  Tolerances: losses 1e-5 relative (CPU) /
1e-4 (GPU, different summation order in the convolutions), parameters after two updates 1e-4 of
their scale."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from conftest import ROOT, load_golden

sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
from make_a2c_golden import FC_STRIDE, init_weights  # noqa: E402  (numpy only; no reference import at load)

import parl_amd as parl  # noqa: E402

NAME = {'fc_pi': 'policy_fc', 'fc_v': 'value_fc'}


def _ours(k):
    head, rest = k.split('.', 1)
    return NAME.get(head, head) + '.' + rest


class TwinModel(parl.Model):
    """stock-torch twin of AtariModel84 (same parameter names), for the CPU host-logic test"""

    def __init__(self, act_dim):
        super().__init__()
        self.conv1 = nn.Conv2d(4, 32, 8, 4, 1)
        self.conv2 = nn.Conv2d(32, 64, 4, 2, 2)
        self.conv3 = nn.Conv2d(64, 64, 3, 1, 0)
        self.fc = nn.Linear(5184, 512)
        self.policy_fc = nn.Linear(512, act_dim)
        self.value_fc = nn.Linear(512, 1)

    def policy_and_value(self, obs):
        x = obs.float() / 255.0
        x = F.relu(self.conv3(F.relu(self.conv2(F.relu(self.conv1(x))))))
        h = F.relu(self.fc(x.flatten(1)))
        return self.policy_fc(h), self.value_fc(h).squeeze(1)

    def policy(self, obs):
        return self.policy_and_value(obs)[0]

    def value(self, obs):
        return self.policy_and_value(obs)[1]


def _check(model, make_alg, dev, rtol_loss, tol_w, lr_frac=0.0):
    z = load_golden('a2c_learn.npz')
    lr_total = float(z['step0/lr_ec'][0] + z['step1/lr_ec'][0])
    A = int(z['dims'][0])
    model.load_state_dict({_ours(k): torch.from_numpy(v) for k, v in init_weights(A).items()})
    model.to(dev)
    alg = make_alg(model)
    t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    p, v = alg.prob_and_value(t(z['step0/obs']))
    np.testing.assert_allclose(p.cpu().numpy(), z['probs0'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(v.cpu().numpy(), z['values0'], rtol=1e-4, atol=1e-5)
    assert np.array_equal(alg.predict(t(z['step0/obs'])).cpu().numpy(), z['predict0'])
    for step in range(2):
        lr, ec = z['step%d/lr_ec' % step]
        out = alg.learn(t(z['step%d/obs' % step]), t(z['step%d/actions' % step]), t(z['step%d/advantages' % step]),
                        t(z['step%d/target_values' % step]), float(lr), float(ec))
        got = np.array([float(x) for x in out])
        np.testing.assert_allclose(got, z['step%d/losses' % step], rtol=rtol_loss, atol=rtol_loss)
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    for k in z:
        if k.startswith('final/'):
            w = sd[_ours(k[6:])]
            # Adam normalises the gradient: after two steps a parameter moved by about lr1 + lr2 whatever the
            # gradient's size, so rounding differences of tiny gradients show up as a fraction of that
            assert np.abs(w - z[k]).max() <= max(tol_w * max(1e-3, np.abs(z[k]).max()), lr_frac * lr_total), k
    w = sd['fc.weight']
    ref = z['final_sample/fc.weight']
    assert np.abs(w.reshape(-1)[::FC_STRIDE] - ref).max() <= max(tol_w * np.abs(ref).max(), lr_frac * lr_total)
    np.testing.assert_allclose([w.astype(np.float64).sum(), np.sqrt((w.astype(np.float64) ** 2).sum())],
                               z['final_stats/fc.weight'], rtol=1e-4 if lr_frac == 0 else 2e-3)  # the plain sum cancels


@pytest.mark.parametrize('style', ['paddle', 'torch'])
def test_a2c_learn_host_logic_matches_reference_torch_a2c(style):
    torch.set_num_threads(4)
    mk = (lambda m: parl.algorithms.A2C(m, vf_loss_coeff=0.5)) if style == 'paddle' else \
        (lambda m: parl.algorithms.A2C(m, {'vf_loss_coeff': 0.5, 'learning_rate': 0.001}))
    _check(TwinModel(6), mk, torch.device('cpu'), 1e-5, 1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize('style', ['paddle', 'torch'])
def test_a2c_learn_on_device_matches_reference_torch_a2c(dev, style):
    from parl_amd.models import AtariModel84
    mk = (lambda m: parl.algorithms.A2C(m, vf_loss_coeff=0.5)) if style == 'paddle' else \
        (lambda m: parl.algorithms.A2C(m, {'vf_loss_coeff': 0.5, 'learning_rate': 0.001}))
    _check(AtariModel84(6), mk, dev, 1e-4, 2e-4, lr_frac=0.1)
