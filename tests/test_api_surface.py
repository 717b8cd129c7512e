"""Host-side mirror of the reference interface: parl.Model / Algorithm / Agent
(parl/core/torch/*), IMPALA / A2C constructors, @parl.remote_class calling convention, utils.
CPU-only; modelled on parl/core/torch/tests/*_test_torch.py and parl/remote/tests/*."""
import os
import threading
import time

import numpy as np
import pytest
import torch
import torch.nn as nn

import parl_amd as parl
from conftest import ROOT
from parl_amd.remote import FutureGetRepeatedlyError, RemoteError


class TinyModel(parl.Model):
    def __init__(self, act_dim=3):
        super(TinyModel, self).__init__()
        self.fc = nn.Linear(4, 8)
        self.pi = nn.Linear(8, act_dim)
        self.v = nn.Linear(8, 1)

    def policy(self, obs):
        return self.pi(torch.relu(self.fc(obs)))

    def value(self, obs):
        return self.v(torch.relu(self.fc(obs))).squeeze(1)

    def policy_and_value(self, obs):
        h = torch.relu(self.fc(obs))
        return self.pi(h), self.v(h).squeeze(1)


def test_model_get_set_weights_roundtrip():
    a, b = TinyModel(), TinyModel()
    w = a.get_weights()
    assert all(isinstance(v, np.ndarray) for v in w.values())  # host numpy copies (model.py:115-123)
    b.set_weights(w)
    x = torch.randn(5, 4)
    assert torch.equal(a.policy(x), b.policy(x))


def test_model_sync_weights_to_decay():
    a, b = TinyModel(), TinyModel()
    wa, wb = a.get_weights(), b.get_weights()
    a.sync_weights_to(b, decay=0.25)
    for k, v in b.get_weights().items():
        if 'weight' in k or 'bias' in k:
            np.testing.assert_allclose(v, 0.25 * wb[k] + 0.75 * wa[k], rtol=1e-6, atol=1e-7)
    with pytest.raises(AssertionError):
        a.sync_weights_to(a)


def test_algorithm_weights_walk():
    class TwoModels(parl.Algorithm):
        def __init__(self, m):
            super(TwoModels, self).__init__(m)
            self.target = TinyModel()
            self.extras = [TinyModel(), 3]

    alg = TwoModels(TinyModel())
    w = alg.get_weights()
    assert set(w) == {'model', 'target', 'extras'} and len(w['extras']) == 1
    alg2 = TwoModels(TinyModel())
    alg2.set_weights(w)
    x = torch.randn(2, 4)
    assert torch.equal(alg.target.policy(x), alg2.target.policy(x))
    with pytest.raises(AssertionError):
        parl.Algorithm(model=nn.Linear(2, 2))  # not a parl.Model


def test_agent_save_restore(tmp_path):
    class A(parl.Agent):
        pass

    alg = parl.algorithms.A2C(TinyModel(), vf_loss_coeff=0.5)
    ag = A(alg)
    p = str(tmp_path / 'sub' / 'model.ckpt')
    ag.save(p)
    alg2 = parl.algorithms.A2C(TinyModel(), vf_loss_coeff=0.5)
    ag2 = A(alg2)
    ag2.restore(p)
    x = torch.randn(3, 4)
    assert torch.equal(alg.model.policy(x), alg2.model.policy(x))
    ag.eval()
    assert not alg.model.training
    ag.train()
    assert alg.model.training


def test_a2c_ctor_both_reference_signatures():
    m = TinyModel()
    a = parl.algorithms.A2C(m, vf_loss_coeff=0.5)  # paddle signature (paddle/a2c.py:26)
    b = parl.algorithms.A2C(m, {'vf_loss_coeff': 0.25, 'learning_rate': 3e-4})  # torch (torch/a2c.py:27)
    c = parl.algorithms.A2C(m, 0.5)
    assert (a.vf_loss_coeff, b.vf_loss_coeff, c.vf_loss_coeff) == (0.5, 0.25, 0.5)
    with pytest.raises(AssertionError):
        parl.algorithms.A2C(nn.Linear(2, 2), vf_loss_coeff=0.5)


def test_a2c_learn_matches_manual_loss_cpu():
    """A2C.learn's loss arithmetic (torch/a2c.py:40-81) is plain torch and runs on CPU."""
    torch.manual_seed(0)
    m = TinyModel()
    alg = parl.algorithms.A2C(m, vf_loss_coeff=0.5)
    obs, act = torch.randn(6, 4), torch.randint(0, 3, (6, ))
    adv, tgt = torch.randn(6), torch.randn(6)
    logits, values = m.policy_and_value(obs)
    logp = torch.log_softmax(logits, 1)
    pi = -(logp.gather(1, act[:, None]).squeeze(1) * adv).sum()
    vf = 0.5 * ((values - tgt)**2).sum()
    ent = -(logp.exp() * logp).sum()
    total, pi_l, vf_l, ent_l = alg.learn(obs, act, adv, tgt, 1e-3, -0.01)
    np.testing.assert_allclose(float(pi_l), float(pi), rtol=1e-6)
    np.testing.assert_allclose(float(vf_l), float(vf), rtol=1e-6)
    np.testing.assert_allclose(float(ent_l), float(ent), rtol=1e-6)
    np.testing.assert_allclose(float(total), float(pi + 0.5 * vf - 0.01 * ent), rtol=1e-6)


def test_impala_ctor_asserts():
    m = TinyModel()
    parl.algorithms.IMPALA(m, sample_batch_steps=5, gamma=0.99, vf_loss_coeff=0.5, clip_rho_threshold=1.0,
                           clip_pg_rho_threshold=1.0)
    with pytest.raises(AssertionError):  # impala.py:100-104: gamma must be a float
        parl.algorithms.IMPALA(m, sample_batch_steps=5, gamma=1, vf_loss_coeff=0.5, clip_rho_threshold=1.0,
                               clip_pg_rho_threshold=1.0)


def test_ops_refuse_cpu_tensors_no_fallback():
    from parl_amd import ops
    from parl_amd._native import ParlHipError
    z = torch.zeros((3, 2))
    with pytest.raises(ParlHipError):
        ops.vtrace(z, z, z, z, z, torch.zeros(2))
    with pytest.raises(ParlHipError):
        ops.gae(z, z, torch.zeros((3, 2), dtype=torch.uint8), torch.zeros(2), 0.99, 0.95)


# ---------------------------------------------------------------- @parl.remote_class
@parl.remote_class
class Counter(object):
    def __init__(self, start=0):
        self.n = start

    def add(self, k):
        self.n += k
        return self.n

    def boom(self):
        raise ValueError('boom')


@parl.remote_class(wait=False, max_memory=300, n_gpu=0)
class SlowCounter(object):
    def __init__(self):
        self.n = 0
        self.thread = None

    def add(self, k):
        time.sleep(0.02)
        self.thread = threading.current_thread().name
        self.n += k
        return self.n

    def boom(self):
        raise ValueError('boom')


def test_remote_class_wait_mode():
    parl.connect('localhost:8010')
    c = Counter(5)
    assert c.add(2) == 7
    assert c.n == 7  # attribute get is proxied (proxy_wrapper.py:69-76)
    c.n = 10
    assert c.add(1) == 11
    with pytest.raises(RemoteError):
        c.boom()


def test_remote_class_nowait_futures():
    c = SlowCounter()
    futs = [c.add(1) for _ in range(5)]
    assert [f.get() for f in futs] == [1, 2, 3, 4, 5]  # one worker thread, call order kept
    assert c.thread != threading.current_thread().name
    with pytest.raises(FutureGetRepeatedlyError):
        futs[0].get()
    with pytest.raises(RemoteError):
        c.boom().get()


def test_remote_class_xparl_env_returns_raw_class(monkeypatch):
    monkeypatch.setenv('XPARL', 'True')  # remote_decorator.py:75-77

    @parl.remote_class(wait=False)
    class Raw(object):
        def f(self):
            return 1

    assert Raw().f() == 1


# ---------------------------------------------------------------- utils
def test_window_and_time_stat():
    w = parl.utils.WindowStat(3)
    for v in (1.0, 2.0, 3.0, 4.0):
        w.add(v)
    assert w.count == 4 and w.mean == 3.0 and w.min == 2.0 and w.max == 4.0
    t = parl.utils.TimeStat(window_size=2)
    with t:
        time.sleep(0.001)
    assert t.mean > 0


# ---- PPO surface (parl/algorithms/torch/ppo.py:27-79, examples/PPO/storage.py:18-43) ----
class TinyPPOModel(parl.Model):
    def __init__(self):
        super(TinyPPOModel, self).__init__()
        self.fc = nn.Linear(4, 8)
        self.pi = nn.Linear(8, 3)
        self.v = nn.Linear(8, 1)

    def policy(self, obs):
        return self.pi(torch.tanh(self.fc(obs)))

    def value(self, obs):
        return self.v(torch.tanh(self.fc(obs)))


def test_ppo_constructor_contract():
    """the argument checks of ppo.py:54-65 and check_model_method (utils.py:217-243)"""
    m = TinyPPOModel()
    alg = parl.algorithms.PPO(m, clip_param=0.2, entropy_coef=0.0, initial_lr=3e-4, continuous_action=False)
    assert alg.clip_param == 0.2 and alg.value_loss_coef == 0.5 and alg.max_grad_norm == 0.5
    assert alg.optimizer.defaults['eps'] == 1e-5 and alg.norm_adv and alg.use_clipped_value_loss
    with pytest.raises(AssertionError):
        parl.algorithms.PPO(m, clip_param=1)              # must be float
    with pytest.raises(AssertionError):
        parl.algorithms.PPO(m, continuous_action=0)       # must be bool
    with pytest.raises(AssertionError):
        parl.algorithms.PPO(TinyModelNoValue())           # model needs value / policy


class TinyModelNoValue(parl.Model):
    def policy(self, obs):
        return obs


def test_rollout_storage_ring_and_loud_failure_without_gpu():
    """append is the reference's ring (cur_step wraps, storage.py:35-43); the arithmetic has no CPU
    path: compute_returns / sample_batch on host tensors raise instead of falling back"""
    import collections
    from parl_amd import _native
    Space = collections.namedtuple('Space', ['shape'])
    rs = parl.RolloutStorage(3, 2, Space((4, )), Space(()), device='cpu')
    assert rs.obs.shape == (3, 2, 4) and rs.actions.shape == (3, 2) and rs.obs.dtype == torch.float32
    for t in range(5):
        rs.append(np.full((2, 4), t, np.float32), np.full(2, t), np.zeros(2), np.ones(2), np.zeros(2), np.zeros(2))
    assert rs.cur_step == 5 % 3
    assert float(rs.obs[0, 0, 0]) == 3.0 and float(rs.obs[1, 0, 0]) == 4.0 and float(rs.obs[2, 0, 0]) == 2.0
    with pytest.raises(_native.ParlHipError):
        rs.compute_returns(np.zeros(2, np.float32), np.zeros(2, np.float32))
    rs.advantages = rs.returns = torch.zeros(3, 2)
    with pytest.raises(_native.ParlHipError):
        rs.sample_batch(np.array([0, 1]))


def test_device_ops_refuse_host_tensors():
    """every new entry fails loudly on CPU tensors (no fallback): conv1-84, vecnorm, fused loss"""
    from parl_amd import _native, ops
    E = _native.ParlHipError
    with pytest.raises(E):
        ops.atari84_conv1(torch.zeros((1, 4, 84, 84), dtype=torch.uint8), torch.zeros(32, 4, 8, 8), torch.zeros(32))
    with pytest.raises(E):
        ops.vecnorm_obs(torch.zeros((2, 3), dtype=torch.float64), torch.zeros((2, 3), dtype=torch.float64),
                        torch.ones((2, 3), dtype=torch.float64), torch.ones(2, dtype=torch.float64))
    with pytest.raises(E):
        ops.impala_loss(torch.zeros(4, 2, 6), torch.zeros(4, 2, 6), torch.zeros((4, 2), dtype=torch.int64),
                        torch.zeros(4, 2), torch.zeros((4, 2), dtype=torch.bool), torch.zeros(4, 2), 0.99)


@pytest.mark.skipif(torch.cuda.is_available(), reason='PPO moves its model to the GPU when one is present')
@pytest.mark.parametrize('case', ['continuous', 'discrete', 'continuous_noclipv_nonorm'])
def test_ppo_learn_host_logic_matches_reference_fixture(case, monkeypatch):
    """parl_amd.algorithms.PPO.learn on CPU against the losses / weights the reference's torch PPO
    produced (tests/golden/ppo_learn.npz).  The advantage normalisation is a HIP kernel with no CPU
    path; for this HOST-LOGIC test it is replaced by the formula it implements (ppo.py:124-127) —
    the kernel itself is checked on the GPU (tests/test_gpu_ppo.py)."""
    from conftest import load_golden
    from parl_amd import ops
    monkeypatch.setattr(ops, 'adv_normalize', lambda adv, eps=1e-8, **k: (adv - adv.mean()) / (adv.std() + eps))
    z = load_golden('ppo_learn.npz')
    obs_dim, act_dim, nb = [int(x) for x in z[case + '/dims']]
    clip, ent, lr0, clipv, norm = [float(x) for x in z[case + '/kw']]
    cont = case.startswith('continuous')

    class MujocoModel(parl.Model):
        def __init__(self):
            super().__init__()
            self.fc1, self.fc2 = nn.Linear(obs_dim, 64), nn.Linear(64, 64)
            self.fc_value, self.fc_policy = nn.Linear(64, 1), nn.Linear(64, act_dim)
            self.fc_pi_std = nn.Parameter(torch.zeros(1, act_dim))

        def value(self, obs):
            return self.fc_value(torch.tanh(self.fc2(torch.tanh(self.fc1(obs)))))

        def policy(self, obs):
            return self.fc_policy(torch.tanh(self.fc2(torch.tanh(self.fc1(obs))))), torch.exp(self.fc_pi_std)

    class DiscreteModel(parl.Model):
        def __init__(self):
            super().__init__()
            self.fc1, self.fc_value, self.fc_policy = nn.Linear(obs_dim, 64), nn.Linear(64, 1), nn.Linear(64, act_dim)

        def value(self, obs):
            return self.fc_value(torch.tanh(self.fc1(obs)))

        def policy(self, obs):
            return self.fc_policy(torch.tanh(self.fc1(obs)))

    model = (MujocoModel if cont else DiscreteModel)()
    prefix = case + '/init/'
    model.load_state_dict({k[len(prefix):]: torch.from_numpy(v) for k, v in z.items() if k.startswith(prefix)})
    alg = parl.algorithms.PPO(model, clip_param=clip, entropy_coef=ent, initial_lr=lr0,
                              use_clipped_value_loss=bool(clipv), norm_adv=bool(norm), continuous_action=cont)
    losses = []
    for it in range(3):
        b = {k: torch.from_numpy(z['%s/batch%d/%s' % (case, it, k)]) for k in ('obs', 'act', 'val', 'ret', 'logp', 'adv')}
        lr = float(z['%s/batch%d/lr' % (case, it)])
        losses.append(alg.learn(b['obs'], b['act'], b['val'], b['ret'], b['logp'], b['adv'], None if np.isnan(lr) else lr))
    np.testing.assert_allclose(np.array(losses), z[case + '/losses'], rtol=1e-5, atol=1e-7)
    prefix = case + '/final/'
    for k, v in alg.model.state_dict().items():
        np.testing.assert_allclose(v.numpy(), z[prefix + k], rtol=1e-4, atol=1e-6, err_msg=k)


def test_compat_paddle_surface_used_by_the_reference_examples():
    """compat/paddle: exactly what examples/IMPALA/*.py and examples/A2C/*.py import from Paddle, on torch"""
    import importlib
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'compat'))
    try:
        for m in [k for k in sys.modules if k == 'paddle' or k.startswith('paddle.')]:
            del sys.modules[m]
        paddle = importlib.import_module('paddle')
        nn = importlib.import_module('paddle.nn')
        F = importlib.import_module('paddle.nn.functional')
    finally:
        sys.path.remove(os.path.join(ROOT, 'compat'))
    assert paddle.__file__.startswith(os.path.join(ROOT, 'compat'))
    x = paddle.to_tensor(np.arange(12, dtype=np.float64).reshape(3, 4), dtype='float32')
    assert isinstance(x, torch.Tensor) and x.dtype == torch.float32 and tuple(x.shape) == (3, 4)
    assert paddle.to_tensor(np.array([1, 0], dtype=np.int32), dtype='int64').dtype == torch.int64
    assert paddle.to_tensor(np.array([True, False]), dtype='bool').dtype == torch.bool
    assert tuple(paddle.squeeze(torch.zeros(5, 1), axis=1).shape) == (5, )
    conv = nn.Conv2D(in_channels=4, out_channels=16, kernel_size=4, stride=2, padding=1)
    fc = nn.Linear(in_features=256, out_features=6,
                   weight_attr=paddle.ParamAttr(initializer=paddle.nn.initializer.Normal()),
                   bias_attr=paddle.ParamAttr(initializer=paddle.nn.initializer.Normal()))
    assert tuple(conv.weight.shape) == (16, 4, 4, 4) and tuple(fc.weight.shape) == (6, 256)
    assert 0.8 < float(fc.weight.std()) < 1.2  # Normal() = N(0, 1), atari_model.py:44-57
    y = F.relu(conv(torch.zeros(2, 4, 42, 42, device=conv.weight.device)))
    assert tuple(y.shape) == (2, 16, 21, 21) and tuple(nn.Flatten()(y).shape) == (2, 16 * 21 * 21)
    loader = paddle.io.DataLoader.from_generator(capacity=5)
    loader.set_batch_generator(lambda: iter([[np.zeros(2)], [np.ones(2)]]))
    assert [float(b[0].sum()) for b in loader()] == [0.0, 2.0]


def test_machine_info():
    from parl_amd.utils import machine_info
    assert machine_info.get_gpu_count() == (torch.cuda.device_count() if torch.cuda.is_available() else 0)
    assert machine_info.is_gpu_available() == (machine_info.get_gpu_count() > 0)
    assert machine_info.is_port_available(int(machine_info.get_free_tcp_port()))
