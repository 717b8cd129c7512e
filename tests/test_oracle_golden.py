"""Pin the CPU oracle (oracle/scan_oracle.c) against the reference's own golden vectors
(tests/golden/*.npz, generated from /root/reference by tests/golden/make_golden.py) and against
numpy's np.random.choice.  CPU-only."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, golden_cases, load_golden


def nan_to_none(x):
    x = float(x)
    return None if np.isnan(x) else x


@pytest.mark.parametrize('case', ['ref_B1', 'ref_B4', 'B7_T13', 'B3_T50', 'B2_T9_noclip'])
def test_vtrace_known_answer(oracle, case):
    g = golden_cases(load_golden('vtrace_known_answer.npz'))[case]
    vs, pg = oracle.vtrace(g['behaviour_actions_log_probs'], g['target_actions_log_probs'], g['discounts'],
                           g['rewards'], g['values'], g['bootstrap_value'],
                           nan_to_none(g['clip_rho_threshold']), nan_to_none(g['clip_pg_rho_threshold']))
    # the reference's own tolerance is 5 decimals (vtrace_test_paddle.py:142-144); values reach
    # ~1e2 here, so also demand 1e-5 relative
    np.testing.assert_allclose(vs, g['vs'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(pg, g['pg_advantages'], rtol=1e-5, atol=1e-5)


def test_vtrace_survey_values(oracle):
    """the numbers recorded in SURVEY.md §8c for the reference recipe"""
    g = golden_cases(load_golden('vtrace_known_answer.npz'))
    b1, b4 = g['ref_B1'], g['ref_B4']
    np.testing.assert_allclose(b1['vs'][:, 0], [0.2001819, 2.7096822, 8.513628, 11.932398, 7.3300004], rtol=1e-6)
    np.testing.assert_allclose(b4['pg_advantages'][:, 3], [0.6683227, 5.3430295, 27.742376, 54.848476, 33.329998],
                               rtol=1e-6)


def test_calc_gae_samples(oracle):
    g = load_golden('calc_gae.npz')
    r = np.array([[1], [0], [-1], [1]], np.float32)
    v = np.array([[.5], [.25], [-.5], [.75]], np.float32)
    d = np.zeros((4, 1), np.uint8)
    adv, _ = oracle.gae(r, v, d, np.array([0.3], np.float32), 0.99, 1.0, accum_f64=True)
    np.testing.assert_allclose(adv[:, 0], g['sample1'], rtol=1e-6)
    d[3] = 1  # next_value = 0 <=> terminal last step
    adv, _ = oracle.gae(r, v, d, np.array([123.0], np.float32), 0.99, 0.95, accum_f64=True)
    np.testing.assert_allclose(adv[:, 0], g['sample2'], rtol=1e-6)


@pytest.mark.parametrize('case', ['a2c_T20_B6_lam1', 'a2c_T20_B6_lam95', 'a2c_T5_B3_lam1', 'a2c_T128_B4_lam9'])
@pytest.mark.parametrize('f64', [True, False])
def test_calc_gae_a2c_segments(oracle, case, f64):
    g = golden_cases(load_golden('calc_gae.npz'))[case]
    adv, ret = oracle.gae(g['rewards'], g['values'], g['dones'], g['next_value'], 0.99, float(g['lam']),
                          accum_f64=f64)
    tol = 2e-6 if f64 else 1e-5
    np.testing.assert_allclose(adv, g['advantages'], rtol=tol, atol=tol)
    np.testing.assert_allclose(ret, g['target_values'], rtol=tol, atol=tol)


def test_discount_sum(oracle):
    g = golden_cases(load_golden('calc_gae.npz'))['dsum']
    out = oracle.discount_cumsum(g['x'], 0.97, accum_f64=True)
    np.testing.assert_allclose(out, g['out'], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('case', ['T16_E8', 'T64_E5', 'T7_E3_g9_l1'])
def test_ppo_compute_returns_bit_exact(oracle, case):
    """examples/PPO/storage.py:45-64 is float32 numpy; the oracle keeps its op order -> bit-exact"""
    g = golden_cases(load_golden('ppo_compute_returns.npz'))[case]
    gamma, lam = [float(x) for x in g['gamma_lam']]
    adv, ret = oracle.gae(g['rewards'], g['values'], g['dones'], g['value'], gamma, lam, last_done=g['done'],
                          done_convention=1)
    assert adv.dtype == np.float32 and g['advantages'].dtype == np.float32
    np.testing.assert_array_equal(adv, g['advantages'])
    np.testing.assert_array_equal(ret, g['returns'])


def test_categorical_matches_np_random_choice(oracle):
    """np.random.choice(len(prob), 1, p=prob) (examples/IMPALA/atari_agent.py:38-40) consumes one
    random_sample() per call; feeding the oracle the same uniforms must give the same indices."""
    rng = np.random.default_rng(7)
    for A in (2, 4, 6, 18):
        logits = rng.standard_normal((500, A)).astype(np.float32) * 2
        e = np.exp(logits - logits.max(1, keepdims=True))
        probs = (e / e.sum(1, keepdims=True)).astype(np.float32)
        np.random.seed(123)
        ref = np.array([np.random.choice(len(p), 1, p=p)[0] for p in probs])
        np.random.seed(123)
        u = np.array([np.random.random_sample() for _ in range(len(probs))])
        got = oracle.categorical_sample(probs, u)
        np.testing.assert_array_equal(got, ref)


def test_categorical_edge_uniforms(oracle):
    probs = np.array([[0.25, 0.25, 0.5], [0.0, 1.0, 0.0], [1.0, 0.0, 0.0]], np.float32)
    for u in (0.0, 0.25, 0.4999999, 0.5, np.nextafter(1.0, 0.0)):
        got = oracle.categorical_sample(probs, np.full(3, u))
        cdf = np.cumsum(probs.astype(np.float64), 1)
        cdf /= cdf[:, -1:]
        ref = np.array([np.searchsorted(c, u, side='right') for c in cdf])
        np.testing.assert_array_equal(got, ref)


def test_philox_known_answer(oracle):
    """Philox4x32-10 known-answer vectors from the Random123 distribution (kat_vectors):
    counter=0,key=0 -> 6627e8d5 e169c58d bc57ac4c 9b00dbd8; all-ones -> 408f276d 41c83b0e a20bc7c6 6d5451fd"""
    import ctypes
    out = (ctypes.c_uint32 * 4)()
    oracle.lib().oracle_philox4x32_10(ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_uint64(0), out)
    assert [hex(x) for x in out] == ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']
    m = 2**64 - 1
    oracle.lib().oracle_philox4x32_10(ctypes.c_uint64(m), ctypes.c_uint64(m), ctypes.c_uint64(m), out)
    assert [hex(x) for x in out] == ['0x408f276d', '0x41c83b0e', '0xa20bc7c6', '0x6d5451fd']


def test_adv_normalize_matches_torch(oracle):
    import torch
    rng = np.random.default_rng(3)
    adv = (rng.standard_normal(4097) * 3 + 1.5).astype(np.float32)
    t = torch.from_numpy(adv)
    ref = ((t - t.mean()) / (t.std() + 1e-8)).numpy()  # torch ppo.py:115-117
    out, ms = oracle.adv_normalize(adv)
    np.testing.assert_allclose(out, ref, rtol=1e-5, atol=1e-6)
    idx = rng.permutation(4097)[:1000]
    ti = t[torch.from_numpy(idx)]
    ref = ((ti - ti.mean()) / (ti.std() + 1e-8)).numpy()
    out, _ = oracle.adv_normalize(adv, idx)
    np.testing.assert_allclose(out, ref, rtol=1e-5, atol=1e-6)


# ---- PPO rows: VecNormalizeEnv running statistics and RolloutStorage.sample_batch ----
def drive_vecnormalize(vn, g, filter_obs, filter_reward):
    """Drive E VecNormalizeEnv twins the way ParallelEnv does (examples/PPO/env_utils.py:62-115):
    reset; then per step: filter obs, filter reward, and for finished envs filter the reset obs."""
    raw, rew, done, rst = g['raw_obs'], g['raw_rew'], g['done'], g['reset_obs']
    S, E, D = raw.shape
    k = np.zeros(E, np.int64)
    first = filter_obs(rst[k, np.arange(E)], None)
    k += 1
    obs, term, r_out = np.zeros((S, E, D)), np.zeros((S, E, D)), np.zeros((S, E))
    for t in range(S):
        o = filter_obs(raw[t], None)
        term[t] = o
        r_out[t] = filter_reward(rew[t], done[t])
        if done[t].any():
            o2 = filter_obs(rst[k, np.arange(E)], done[t].astype(np.uint8))
            o = np.where(done[t][:, None], o2, o)
            k += done[t]
        obs[t] = o
    return first, obs, term, r_out


@pytest.mark.parametrize('case', ['E6_D17_S60', 'E3_D5_S200'])
def test_vecnormalize_bit_exact(oracle, case):
    """oracle/ppo_oracle.c == the reference's VecNormalizeEnv / RunningMeanStd (float64, bit-exact)"""
    g = golden_cases(load_golden('vecnormalize.npz'))[case]
    S, E, D = g['raw_obs'].shape
    vn = oracle.VecNormalize(E, D, gamma=0.99)
    first, obs, term, rew = drive_vecnormalize(vn, g, vn.filter_obs, vn.filter_reward)
    assert np.array_equal(first, g['first_obs'])
    assert np.array_equal(term, g['obs_terminal'])
    assert np.array_equal(obs, g['obs'])
    assert np.array_equal(rew, g['rew'])
    for mine, ref in [(vn.ob_mean, 'ob_mean'), (vn.ob_var, 'ob_var'), (vn.ob_count, 'ob_count'),
                      (vn.ret_mean, 'ret_mean'), (vn.ret_var, 'ret_var'), (vn.ret_count, 'ret_count'),
                      (vn.ret, 'ret')]:
        assert np.array_equal(mine, g[ref]), ref


@pytest.mark.parametrize('case', ['T12_E6', 'T9_E4_discrete'])
def test_ppo_sample_batch_oracle(oracle, case):
    """ring append (storage.py:35-43) + compute_returns + sample_batch (storage.py:66-76)"""
    g = golden_cases(load_golden('ppo_sample_batch.npz'))[case]
    steps, E = g['append_rewards'].shape
    T = steps - 5
    ring = {k: np.zeros((T, ) + g['append_' + k].shape[1:], np.float32)
            for k in ('obs', 'actions', 'logprobs', 'rewards', 'dones', 'values')}
    cur = 0
    for t in range(steps):
        for k in ring:
            ring[k][cur] = g['append_' + k][t]
        cur = (cur + 1) % T
    assert cur == int(g['cur_step'])
    adv, ret = oracle.gae(ring['rewards'], ring['values'], ring['dones'], g['value'], 0.99, 0.95,
                          last_done=g['done'], done_convention=1)
    out = oracle.ppo_sample_batch(ring['obs'], ring['actions'], ring['logprobs'], adv, ret, ring['values'], g['idx'])
    for o, k in zip(out, ['obs', 'actions', 'logprobs', 'advantages', 'returns', 'values']):
        assert np.array_equal(o.reshape(g['batch_' + k].shape), g['batch_' + k]), k


# ---- the numpy CPU baselines bench.py times (oracle/py_baselines.py) are pinned on the same fixtures ----
def test_py_baselines_match_reference_fixtures():
    from oracle import py_baselines as pb
    z = load_golden('calc_gae.npz')
    for case in ('a2c_T20_B6_lam1', 'a2c_T20_B6_lam95'):
        c = {k.split('/', 1)[1]: v for k, v in z.items() if k.startswith(case + '/')}
        adv = pb.calc_gae_segments(c['rewards'], c['values'], c['dones'].astype(bool), c['next_value'], 0.99,
                                   float(c['lam']))
        np.testing.assert_allclose(adv, c['advantages'], rtol=1e-6, atol=1e-6)
    z = load_golden('ppo_compute_returns.npz')
    for case in ('T16_E8', 'T64_E5'):
        c = {k.split('/', 1)[1]: v for k, v in z.items() if k.startswith(case + '/')}
        g, lam = c['gamma_lam']
        adv, ret = pb.compute_returns(c['rewards'], c['values'], c['dones'], c['value'], c['done'], float(g), float(lam))
        assert np.array_equal(adv, c['advantages']) and np.array_equal(ret, c['returns'])
    z = load_golden('vtrace_known_answer.npz')
    for case in ('ref_B1', 'ref_B4'):
        c = {k.split('/', 1)[1]: v for k, v in z.items() if k.startswith(case + '/')}
        vs, pg = pb.vtrace_numpy(c['behaviour_actions_log_probs'], c['target_actions_log_probs'], c['discounts'],
                                 c['rewards'], c['values'], c['bootstrap_value'], float(c['clip_rho_threshold']),
                                 float(c['clip_pg_rho_threshold']))
        np.testing.assert_allclose(vs, c['vs'], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(pg, c['pg_advantages'], rtol=1e-5, atol=1e-5)


def test_torch_cpu_baselines_match_reference_fixtures():
    """oracle/ref_torch_baselines.py (bench.py's torch-CPU baselines): the torch port of the
    reference's per-t V-trace loop reproduces the reference's known-answer vectors; the staged reference torch
    A2C.learn + ActorCritic (where build() staged them) load by path and reproduce the losses the same reference
    classes produced for tests/golden/a2c_learn.npz"""
    import torch
    from oracle import ref_torch_baselines as rb
    z = load_golden('vtrace_known_answer.npz')
    for case in ('ref_B1', 'ref_B4'):
        c = {k.split('/', 1)[1]: v for k, v in z.items() if k.startswith(case + '/')}
        t = lambda k: torch.from_numpy(np.ascontiguousarray(c[k]))  # noqa: E731
        vs, pg = rb.vtrace_torch(t('behaviour_actions_log_probs'), t('target_actions_log_probs'), t('discounts'),
                                 t('rewards'), t('values'), t('bootstrap_value'), float(c['clip_rho_threshold']),
                                 float(c['clip_pg_rho_threshold']))
        np.testing.assert_allclose(vs.numpy(), c['vs'], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(pg.numpy(), c['pg_advantages'], rtol=1e-5, atol=1e-5)
    if not os.path.exists(os.path.join(rb.STAGED, 'a2c.py')):
        pytest.skip('oracle/_ref/torch_alg not staged (no /root/reference at build time)')
    A2C, ActorCritic = rb._load_reference_a2c()
    assert getattr(sys.modules.get('parl'), 'Model', None) is not torch.nn.Module  # the import stub is gone again
    g = load_golden('a2c_learn.npz')
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    try:
        from make_a2c_golden import init_weights
    finally:
        sys.path.pop(0)
    A = int(g['dims'][0])
    torch.manual_seed(3)
    n0 = torch.get_num_threads()
    torch.set_num_threads(4)
    try:
        model = ActorCritic(A)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in init_weights(A).items()})
        alg = A2C(model, {'vf_loss_coeff': 0.5, 'learning_rate': 0.001})
        out = alg.learn(torch.from_numpy(g['step0/obs']).float(), torch.from_numpy(g['step0/actions']),
                        torch.from_numpy(g['step0/advantages']), torch.from_numpy(g['step0/target_values']), 1e-3, -0.01)
    finally:
        torch.set_num_threads(n0)
    np.testing.assert_allclose([float(x.detach()) for x in out], g['step0/losses'], rtol=1e-4)
