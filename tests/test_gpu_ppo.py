"""PPO rows of SURVEY.md §8 (a10, f4) on the GPU, through the C ABI, against fixtures generated
from the reference's own classes (tests/golden/make_ppo_golden.py):
  * DeviceVecNormalize == VecNormalizeEnv / RunningMeanStd (parl/env/mujoco_wrappers.py:73-206),
    float64 statistics and outputs bit-exact, float32 rows == the numpy cast,
  * RolloutStorage (append ring, compute_returns, sample_batch) == examples/PPO/storage.py, bit-exact,
  * PPO.learn == parl/algorithms/torch/ppo.py on CPU, losses to 1e-4 relative (fp32 GEMM order).
-m gpu."""
import collections

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import golden_cases, load_golden
from test_oracle_golden import drive_vecnormalize

pytestmark = pytest.mark.gpu
Space = collections.namedtuple('Space', ['shape'])


@pytest.mark.parametrize('case', ['E6_D17_S60', 'E3_D5_S200'])
def test_device_vecnormalize_bit_exact(dev, case):
    from parl_amd import ops
    from parl_amd.env import DeviceVecNormalize
    g = golden_cases(load_golden('vecnormalize.npz'))[case]
    S, E, D = g['raw_obs'].shape
    vn = DeviceVecNormalize(E, D, gamma=0.99, device=dev)
    f32_rows = []

    def filter_obs(raw, mask):
        raw = torch.from_numpy(np.ascontiguousarray(raw)).to(dev)
        o64 = torch.zeros((E, D), dtype=torch.float64, device=dev)
        m = None if mask is None else torch.from_numpy(mask).to(dev)
        o32 = ops.vecnorm_obs(raw, vn.ob_mean, vn.ob_var, vn.ob_count, mask=m, out64=o64, clipob=vn.clipob,
                              eps=vn.epsilon, update=True)
        f32_rows.append((o32.cpu().numpy(), o64.cpu().numpy(), mask))
        return o64.cpu().numpy()

    def filter_reward(rew, done):
        o64 = torch.zeros(E, dtype=torch.float64, device=dev)
        o32 = ops.vecnorm_reward(torch.from_numpy(np.ascontiguousarray(rew)).to(dev), torch.from_numpy(done).to(dev),
                                 vn.ret, vn.ret_mean, vn.ret_var, vn.ret_count, gamma=0.99, out64=o64)
        assert np.array_equal(o32.cpu().numpy(), o64.cpu().numpy().astype(np.float32))
        return o64.cpu().numpy()

    first, obs, term, rew = drive_vecnormalize(vn, g, filter_obs, filter_reward)
    assert np.array_equal(first, g['first_obs'])
    assert np.array_equal(term, g['obs_terminal'])
    assert np.array_equal(obs, g['obs'])
    assert np.array_equal(rew, g['rew'])
    for o32, o64, mask in f32_rows:  # the float32 rows are the numpy cast of the float64 result
        sel = slice(None) if mask is None else mask.astype(bool)
        assert np.array_equal(o32[sel], o64[sel].astype(np.float32))
    for mine, ref in [(vn.ob_mean, 'ob_mean'), (vn.ob_var, 'ob_var'), (vn.ob_count, 'ob_count'),
                      (vn.ret_mean, 'ret_mean'), (vn.ret_var, 'ret_var'), (vn.ret_count, 'ret_count'),
                      (vn.ret, 'ret')]:
        assert np.array_equal(mine.cpu().numpy(), g[ref]), ref


def test_device_vecnormalize_class_flow(dev):
    """the host mirror (reset / step / reset_where, eval mode) against the same golden stream"""
    from parl_amd.env import DeviceVecNormalize
    g = golden_cases(load_golden('vecnormalize.npz'))['E6_D17_S60']
    raw, rew, done, rst = g['raw_obs'], g['raw_rew'], g['done'], g['reset_obs']
    S, E, D = raw.shape
    vn = DeviceVecNormalize(E, D, gamma=0.99, device=dev)
    k = np.zeros(E, np.int64)
    first = vn.reset(rst[k, np.arange(E)])
    k += 1
    assert np.array_equal(first.cpu().numpy(), g['first_obs'].astype(np.float32))
    for t in range(S):
        o, r = vn.step(raw[t], rew[t], done[t])
        if done[t].any():
            o = vn.reset_where(done[t], rst[k, np.arange(E)], o)
            k += done[t]
        assert np.array_equal(o.cpu().numpy(), g['obs'][t].astype(np.float32)), t
        assert np.array_equal(r.cpu().numpy(), g['rew'][t].astype(np.float32)), t
    # eval mode: statistics frozen (VecNormalizeEnv.eval, mujoco_wrappers.py:166-167)
    vn.eval()
    before = vn.ob_mean.clone()
    o = vn.reset(raw[0])
    assert torch.equal(before, vn.ob_mean)
    want = np.clip((raw[0] - g['ob_mean']) / np.sqrt(g['ob_var'] + 1e-8), -10, 10).astype(np.float32)
    assert np.array_equal(o.cpu().numpy(), want)
    rms = vn.get_ob_rms()
    vn2 = DeviceVecNormalize(E, D, device=dev)
    vn2.set_ob_rms(rms)
    assert torch.equal(vn2.ob_var, vn.ob_var) and torch.equal(vn2.ob_count, vn.ob_count)


@pytest.mark.parametrize('case', ['T12_E6', 'T9_E4_discrete'])
def test_rollout_storage_matches_reference(dev, case):
    import parl_amd as parl
    g = golden_cases(load_golden('ppo_sample_batch.npz'))[case]
    steps, E = g['append_rewards'].shape
    T = steps - 5
    rs = parl.RolloutStorage(T, E, Space(g['append_obs'].shape[2:]), Space(g['append_actions'].shape[2:]), device=dev)
    for t in range(steps):
        row = [g['append_' + k][t] for k in ('obs', 'actions', 'logprobs', 'rewards', 'dones', 'values')]
        if t % 2:  # device rows are accepted as well as host rows
            row = [torch.from_numpy(x).to(dev) for x in row]
        rs.append(*row)
    assert rs.cur_step == int(g['cur_step'])
    rs.compute_returns(g['value'], g['done'])
    out = rs.sample_batch(g['idx'])
    for o, k in zip(out, ['obs', 'actions', 'logprobs', 'advantages', 'returns', 'values']):
        assert np.array_equal(o.cpu().numpy(), g['batch_' + k]), k
    from parl_amd import ops
    assert ops.consume_device_errors() == 0
    bad = g['idx'].copy()
    bad[0] = T * E  # numpy raises IndexError; the kernel flags it
    rs.sample_batch(bad)
    assert ops.consume_device_errors() > 0


@pytest.mark.parametrize('case', ['T16_E8', 'T64_E5', 'T7_E3_g9_l1'])
def test_rollout_storage_compute_returns_bit_exact(dev, case):
    import parl_amd as parl
    g = golden_cases(load_golden('ppo_compute_returns.npz'))[case]
    T, E = g['rewards'].shape
    rs = parl.RolloutStorage(T, E, Space((3, )), Space((2, )), device=dev)
    rs.rewards.copy_(torch.from_numpy(g['rewards']))
    rs.values.copy_(torch.from_numpy(g['values']))
    rs.dones.copy_(torch.from_numpy(g['dones']))
    adv, ret = rs.compute_returns(g['value'], g['done'], gamma=float(g['gamma_lam'][0]),
                                  gae_lambda=float(g['gamma_lam'][1]))
    assert np.array_equal(adv.cpu().numpy(), g['advantages'])
    assert np.array_equal(ret.cpu().numpy(), g['returns'])


def test_storage_c5_shape_chunked_vs_single_pass(dev):
    """BASELINE configs[4] shape (T=2048, E=4096): the chunk-parallel plan agrees with the bit-exact
    single pass to 1e-5, and minibatch normalisation of gathered advantages has mean 0 / std 1"""
    import parl_amd as parl
    from parl_amd import ops
    T, E = 2048, 4096
    gen = torch.Generator(device=dev).manual_seed(0)
    rs = parl.RolloutStorage(T, E, Space((17, )), Space((6, )), device=dev)
    rs.rewards.copy_(torch.randn((T, E), device=dev, generator=gen).clamp_(-10, 10))
    rs.values.copy_(torch.randn((T, E), device=dev, generator=gen))
    phase = torch.randint(0, 1000, (E, ), device=dev, generator=gen)
    rs.dones.copy_((((torch.arange(T, device=dev)[:, None] + phase[None]) % 1000) == 0).float())
    nv, ld = torch.randn(E, device=dev, generator=gen), torch.zeros(E, device=dev)
    a1, r1 = rs.compute_returns(nv, ld)
    a1, r1 = a1.clone(), r1.clone()
    a2, r2 = rs.compute_returns(nv, ld, chunked=True)
    np.testing.assert_allclose(a2.cpu().numpy(), a1.cpu().numpy(), rtol=1e-5, atol=1e-5)
    idx = torch.randperm(T * E, device=dev, generator=gen)[:T * E // 32]
    b = rs.sample_batch(idx)
    assert torch.equal(b[3], a2.reshape(-1)[idx]) and torch.equal(b[0], rs.obs.reshape(-1, 17)[idx])
    nrm = ops.adv_normalize(b[3])
    assert abs(float(nrm.mean())) < 1e-5 and abs(float(nrm.std()) - 1.0) < 1e-4


class _MujocoModel(nn.Module):
    pass


def _models():
    import parl_amd as parl

    class MujocoModel(parl.Model):  # examples/PPO/mujoco_model.py:21-63 (torch twin)
        def __init__(self, obs_dim, act_dim):
            super().__init__()
            self.fc1 = nn.Linear(obs_dim, 64)
            self.fc2 = nn.Linear(64, 64)
            self.fc_value = nn.Linear(64, 1)
            self.fc_policy = nn.Linear(64, act_dim)
            self.fc_pi_std = nn.Parameter(torch.zeros(1, act_dim))

        def value(self, obs):
            return self.fc_value(torch.tanh(self.fc2(torch.tanh(self.fc1(obs)))))

        def policy(self, obs):
            out = torch.tanh(self.fc2(torch.tanh(self.fc1(obs))))
            return self.fc_policy(out), torch.exp(self.fc_pi_std)

    class DiscreteModel(parl.Model):
        def __init__(self, obs_dim, act_dim):
            super().__init__()
            self.fc1 = nn.Linear(obs_dim, 64)
            self.fc_value = nn.Linear(64, 1)
            self.fc_policy = nn.Linear(64, act_dim)

        def value(self, obs):
            return self.fc_value(torch.tanh(self.fc1(obs)))

        def policy(self, obs):
            return self.fc_policy(torch.tanh(self.fc1(obs)))

    return MujocoModel, DiscreteModel


@pytest.mark.parametrize('case', ['continuous', 'discrete', 'continuous_noclipv_nonorm'])
def test_ppo_learn_matches_reference(dev, case):
    import parl_amd as parl
    MujocoModel, DiscreteModel = _models()
    g = golden_cases(load_golden('ppo_learn.npz'))
    z = load_golden('ppo_learn.npz')
    obs_dim, act_dim, nb = [int(x) for x in z[case + '/dims']]
    clip, ent, lr0, clipv, norm = [float(x) for x in z[case + '/kw']]
    cont = case.startswith('continuous')
    model = (MujocoModel if cont else DiscreteModel)(obs_dim, act_dim)
    prefix = case + '/init/'
    model.load_state_dict({k[len(prefix):]: torch.from_numpy(v) for k, v in z.items() if k.startswith(prefix)})
    alg = parl.algorithms.PPO(model, clip_param=clip, entropy_coef=ent, initial_lr=lr0,
                              use_clipped_value_loss=bool(clipv), norm_adv=bool(norm), continuous_action=cont)
    assert next(alg.model.parameters()).is_cuda
    losses = []
    for it in range(3):
        b = {k: torch.from_numpy(z['%s/batch%d/%s' % (case, it, k)]).to(dev) for k in ('obs', 'act', 'val', 'ret', 'logp', 'adv')}
        lr = float(z['%s/batch%d/lr' % (case, it)])
        losses.append(alg.learn(b['obs'], b['act'], b['val'], b['ret'], b['logp'], b['adv'], None if np.isnan(lr) else lr))
    np.testing.assert_allclose(np.array(losses), z[case + '/losses'], rtol=1e-4, atol=1e-6)
    prefix = case + '/final/'
    for k, v in alg.model.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), z[prefix + k], rtol=1e-3, atol=5e-5, err_msg=k)
    # sample / predict / value contracts (ppo.py:160-206)
    o = torch.randn(9, obs_dim, device=dev)
    value, action, logp, entropy = alg.sample(o)
    assert value.shape == (9, 1) and logp.shape == (9, ) and entropy.shape == (9, )
    assert action.shape == ((9, act_dim) if cont else (9, ))
    assert alg.predict(o).shape == ((9, act_dim) if cont else (9, 1))
    assert alg.value(o).shape == (9, 1)
