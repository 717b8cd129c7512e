"""Parity of the HIP scan / sampling kernels (through the C ABI) against the CPU oracle and
the reference's golden vectors.  All tests need a real MI355X: run with -m gpu."""
import numpy as np
import pytest
import torch

from conftest import golden_cases, load_golden

pytestmark = pytest.mark.gpu

# tolerance stated by BASELINE.json north_star: returns / V-trace targets to 1e-5 fp32
RTOL = 1e-5
ATOL = 1e-5


def T_(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def nan_to_none(x):
    x = float(x)
    return None if np.isnan(x) else x


def vtrace_inputs(rng, T, B):
    blp = -np.abs(rng.standard_normal((T, B))).astype(np.float32)
    tlp = -np.abs(rng.standard_normal((T, B))).astype(np.float32)
    dones = rng.random((T, B)) < 1 / 50
    disc = (~dones).astype(np.float32) * np.float32(0.99)
    rew = rng.choice([-1.0, 0.0, 1.0], size=(T, B), p=[.01, .98, .01]).astype(np.float32)
    val = rng.standard_normal((T, B)).astype(np.float32)
    boot = rng.standard_normal(B).astype(np.float32)
    return blp, tlp, disc, rew, val, boot


@pytest.mark.parametrize('case', ['ref_B1', 'ref_B4', 'B7_T13', 'B3_T50', 'B2_T9_noclip'])
def test_vtrace_golden(dev, case):
    from parl_amd import ops
    g = golden_cases(load_golden('vtrace_known_answer.npz'))[case]
    vs, pg = ops.vtrace(T_(g['behaviour_actions_log_probs'], dev), T_(g['target_actions_log_probs'], dev),
                        T_(g['discounts'], dev), T_(g['rewards'], dev), T_(g['values'], dev),
                        T_(g['bootstrap_value'], dev), nan_to_none(g['clip_rho_threshold']),
                        nan_to_none(g['clip_pg_rho_threshold']))
    np.testing.assert_allclose(vs.cpu().numpy(), g['vs'], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(pg.cpu().numpy(), g['pg_advantages'], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize('T,B', [(49, 1024), (49, 20), (1, 1), (5, 3), (127, 4099), (19, 131072 + 4), (7, 262144)])
@pytest.mark.parametrize('clips', [(1.0, 1.0), (3.7, 2.2), (None, None)])
def test_vtrace_vs_oracle(dev, oracle, T, B, clips):
    from parl_amd import ops
    rng = np.random.default_rng(T * 1000 + B)
    inp = vtrace_inputs(rng, T, B)
    vs, pg = ops.vtrace(*[T_(x, dev) for x in inp], clips[0], clips[1])
    ovs, opg = oracle.vtrace(*inp, clips[0], clips[1])
    np.testing.assert_allclose(vs.cpu().numpy(), ovs, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(pg.cpu().numpy(), opg, rtol=RTOL, atol=ATOL)


def test_vtrace_empty(dev):
    from parl_amd import ops
    z = torch.zeros((0, 4), device=dev)
    vs, pg = ops.vtrace(z, z, z, z, z, torch.zeros(4, device=dev))
    assert vs.shape == (0, 4)


def test_vtrace_full_size_properties(dev):
    """saturating shape (T'=127, B=262,144): on-policy (rho=1), no clipping effect,
    V-trace reduces to the n-step TD(1) return — checked against a float64 torch scan on device,
    plus linearity in (rewards, values, bootstrap)."""
    from parl_amd import ops
    T, B = 127, 262144
    g = torch.Generator(device=dev).manual_seed(0)
    lp = -torch.rand((T, B), device=dev, generator=g)
    disc = (torch.rand((T, B), device=dev, generator=g) > 1 / 800).float() * 0.99
    rew = torch.randn((T, B), device=dev, generator=g)
    val = torch.randn((T, B), device=dev, generator=g)
    boot = torch.randn(B, device=dev, generator=g)
    vs, pg = ops.vtrace(lp, lp, disc, rew, val, boot, 1.0, 1.0)
    # reference: vs_t = r_t + disc_t * vs_{t+1}, vs_T = bootstrap (float64)
    ref = torch.empty((T, B), device=dev, dtype=torch.float64)
    nxt = boot.double()
    for t in range(T - 1, -1, -1):
        nxt = rew[t].double() + disc[t].double() * nxt
        ref[t] = nxt
    torch.testing.assert_close(vs.double(), ref, rtol=1e-5, atol=1e-4)
    nvs = torch.cat([vs[1:], boot[None]], 0)
    torch.testing.assert_close(pg, rew + disc * nvs - val, rtol=1e-5, atol=1e-5)
    # linearity: doubling rewards, values, bootstrap doubles vs and pg
    vs2, pg2 = ops.vtrace(lp, lp, disc, rew * 2, val * 2, boot * 2, 1.0, 1.0)
    torch.testing.assert_close(vs2, vs * 2, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(pg2, pg * 2, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('time_major', [True, False])
@pytest.mark.parametrize('T,B,A', [(50, 1024, 6), (50, 20, 6), (2, 1, 4), (20, 257, 4), (9, 70, 18), (65, 33, 6),
                                   (66, 5, 3), (130, 7, 2), (300, 3, 7), (50, 16384, 6)])
def test_vtrace_from_logits_vs_oracle(dev, oracle, time_major, T, B, A):
    from parl_amd import ops
    rng = np.random.default_rng(T + B + A)
    shp = (T, B) if time_major else (B, T)
    bl = rng.standard_normal(shp + (A, )).astype(np.float32)
    tl = (bl + 0.3 * rng.standard_normal(shp + (A, ))).astype(np.float32)
    act = rng.integers(0, A, shp).astype(np.int64)
    rew = rng.choice([-1.0, 0.0, 1.0], size=shp, p=[.05, .9, .05]).astype(np.float32)
    dones = rng.random(shp) < 0.03
    val = rng.standard_normal(shp).astype(np.float32)
    vs, pg, tlp, blp = ops.vtrace_from_logits(T_(bl, dev), T_(tl, dev), T_(act, dev), T_(rew, dev), T_(dones, dev),
                                              T_(val, dev), 0.99, 1.0, 1.0, time_major=time_major,
                                              want_log_probs=True)
    ovs, opg, otlp, oblp = oracle.vtrace_from_logits(bl, tl, act, rew, dones, val, 0.99, 1.0, 1.0, time_major)
    np.testing.assert_allclose(tlp.cpu().numpy(), otlp, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(blp.cpu().numpy(), oblp, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(vs.cpu().numpy(), ovs, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(pg.cpu().numpy(), opg, rtol=RTOL, atol=ATOL)
    assert ops.consume_device_errors() == 0


def test_vtrace_from_logits_bad_action_flag(dev):
    from parl_amd import ops
    T, B, A = 5, 4, 6
    z = torch.zeros((T, B, A), device=dev)
    act = torch.full((T, B), 9, dtype=torch.int64, device=dev)
    s = torch.zeros((T, B), device=dev)
    ops.vtrace_from_logits(z, z, act, s, s.bool(), s, 0.99)
    assert ops.consume_device_errors() == 1
    assert ops.consume_device_errors() == 0


@pytest.mark.parametrize('case', ['a2c_T20_B6_lam1', 'a2c_T20_B6_lam95', 'a2c_T5_B3_lam1', 'a2c_T128_B4_lam9'])
def test_gae_a2c_golden(dev, case):
    """HIP float32 scan vs the reference calc_gae (float64 lfilter) per (env, segment)"""
    from parl_amd import ops
    g = golden_cases(load_golden('calc_gae.npz'))[case]
    adv, ret = ops.gae(T_(g['rewards'], dev), T_(g['values'], dev), T_(g['dones'], dev), T_(g['next_value'], dev),
                       0.99, float(g['lam']))
    np.testing.assert_allclose(adv.cpu().numpy(), g['advantages'], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(ret.cpu().numpy(), g['target_values'], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize('case', ['T16_E8', 'T64_E5', 'T7_E3_g9_l1'])
def test_gae_ppo_golden_bit_exact(dev, case):
    """RolloutStorage.compute_returns is float32 numpy: the kernel keeps its op order -> bit-exact"""
    from parl_amd import ops
    g = golden_cases(load_golden('ppo_compute_returns.npz'))[case]
    gamma, lam = [float(x) for x in g['gamma_lam']]
    adv, ret = ops.gae(T_(g['rewards'], dev), T_(g['values'], dev), T_(g['dones'], dev), T_(g['value'], dev), gamma,
                       lam, last_done=T_(g['done'], dev), done_convention=ops.GAE_DONE_STARTS_STEP)
    np.testing.assert_array_equal(adv.cpu().numpy(), g['advantages'])
    np.testing.assert_array_equal(ret.cpu().numpy(), g['returns'])


@pytest.mark.parametrize('T,B', [(20, 256), (20, 5), (1, 1), (128, 1031), (33, 131072), (2048, 4096), (300, 77),
                                 (1000, 5000)])
@pytest.mark.parametrize('conv,f32', [(0, False), (1, True), (1, False), (0, True)])
def test_gae_vs_oracle(dev, oracle, T, B, conv, f32):
    from parl_amd import ops
    rng = np.random.default_rng(T * 7 + B)
    rew = np.clip(rng.standard_normal((T, B)), -10, 10).astype(np.float32)
    val = rng.standard_normal((T, B)).astype(np.float32)
    dones = (rng.random((T, B)) < 0.02)
    last = (rng.random(B) < 0.3)
    dt = np.float32 if f32 else np.uint8
    dones, last = dones.astype(dt), last.astype(dt)
    nv = rng.standard_normal(B).astype(np.float32)
    adv, ret = ops.gae(T_(rew, dev), T_(val, dev), T_(dones, dev), T_(nv, dev), 0.99, 0.95,
                       last_done=T_(last, dev) if conv else None, done_convention=conv)
    oadv, oret = oracle.gae(rew, val, dones, nv, 0.99, 0.95, last_done=last if conv else None, done_convention=conv)
    from parl_amd import _native as N
    chunked = N.lib().parlhip_gae_workspace_bytes(T, B) > 0
    if conv == 1 and chunked:
        # chunk-parallel plan (long T, few sequences): fp32 re-association of the carry only;
        # the single-pass kernel keeps numpy's op order bit-exactly on the same inputs
        a1, r1 = ops.gae(T_(rew, dev), T_(val, dev), T_(dones, dev), T_(nv, dev), 0.99, 0.95,
                         last_done=T_(last, dev), done_convention=conv, allow_chunked=False)
        np.testing.assert_array_equal(a1.cpu().numpy(), oadv)
        np.testing.assert_array_equal(r1.cpu().numpy(), oret)
    if conv == 1 and not chunked:  # same float32 op order on both sides
        np.testing.assert_array_equal(adv.cpu().numpy(), oadv)
        np.testing.assert_array_equal(ret.cpu().numpy(), oret)
    else:
        np.testing.assert_allclose(adv.cpu().numpy(), oadv, rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(ret.cpu().numpy(), oret, rtol=RTOL, atol=ATOL)


def test_gae_fp32_vs_reference_f64_T2048(dev, oracle):
    """SURVEY hard part: fp32 scan vs the reference's float64 lfilter at gamma=.99, lam=1, T=2048"""
    from parl_amd import ops
    rng = np.random.default_rng(5)
    T, B = 2048, 64
    rew = rng.choice([-1.0, 0.0, 1.0], size=(T, B), p=[.01, .98, .01]).astype(np.float32)
    val = rng.standard_normal((T, B)).astype(np.float32)
    dones = np.zeros((T, B), np.uint8)
    nv = rng.standard_normal(B).astype(np.float32)
    adv, _ = ops.gae(T_(rew, dev), T_(val, dev), T_(dones, dev), T_(nv, dev), 0.99, 1.0)
    oadv, _ = oracle.gae(rew, val, dones, nv, 0.99, 1.0, accum_f64=True)
    np.testing.assert_allclose(adv.cpu().numpy(), oadv, rtol=1e-5, atol=2e-5)


def test_discount_cumsum(dev, oracle):
    from parl_amd import ops
    g = golden_cases(load_golden('calc_gae.npz'))['dsum']
    out = ops.discount_cumsum(T_(g['x'], dev), 0.97)
    np.testing.assert_allclose(out.cpu().numpy(), g['out'], rtol=RTOL, atol=ATOL)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((40, 70000)).astype(np.float32)
    d = rng.random((40, 70000)) < 0.05
    out = ops.discount_cumsum(T_(x, dev), 0.9, T_(d, dev))
    np.testing.assert_allclose(out.cpu().numpy(), oracle.discount_cumsum(x, 0.9, d), rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize('n', [1000, 262144, 4096 * 2048 // 32, 3])
def test_adv_normalize(dev, oracle, n):
    from parl_amd import ops
    rng = np.random.default_rng(n)
    adv = (rng.standard_normal(n * 2) * 2 + 0.7).astype(np.float32)
    out, st = ops.adv_normalize(T_(adv, dev), return_stats=True)
    oout, oms = oracle.adv_normalize(adv)
    np.testing.assert_allclose(st.cpu().numpy(), oms, rtol=1e-6)
    np.testing.assert_allclose(out.cpu().numpy(), oout, rtol=RTOL, atol=ATOL)
    idx = rng.permutation(n * 2)[:n].astype(np.int64)
    out = ops.adv_normalize(T_(adv, dev), T_(idx, dev))
    oout, _ = oracle.adv_normalize(adv, idx)
    np.testing.assert_allclose(out.cpu().numpy(), oout, rtol=RTOL, atol=ATOL)
    # property at any size: zero mean, unit (unbiased) std
    if n > 100:
        assert abs(float(out.mean())) < 1e-4 and abs(float(out.std()) - 1) < 1e-4


@pytest.mark.parametrize('A', [2, 4, 6, 18, 40])
def test_categorical_sample_bit_exact(dev, oracle, A):
    from parl_amd import ops
    rng = np.random.default_rng(A)
    B = 5000
    logits = rng.standard_normal((B, A)).astype(np.float32) * 2
    e = np.exp(logits - logits.max(1, keepdims=True))
    probs = (e / e.sum(1, keepdims=True)).astype(np.float32)
    u = rng.random(B)
    u[:4] = [0.0, np.nextafter(1.0, 0.0), 0.5, 0.25]
    got = ops.categorical_sample(T_(probs, dev), T_(u, dev)).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.categorical_sample(probs, u))
    # and against numpy's own choice on the same uniform stream
    np.random.seed(11)
    ref = np.array([np.random.choice(A, 1, p=p)[0] for p in probs[:300]])
    np.random.seed(11)
    uu = np.array([np.random.random_sample() for _ in range(300)])
    got = ops.categorical_sample(T_(probs[:300], dev), T_(uu, dev)).cpu().numpy()
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize('A', [4, 6, 18])
def test_policy_sample(dev, oracle, A):
    from parl_amd import ops
    rng = np.random.default_rng(A + 100)
    B = 4096
    logits = (rng.standard_normal((B, A)) * 3).astype(np.float32)
    seed, offset, row0 = 0x1234567890abcdef, 77, 1 << 33
    act, probs, uni = ops.policy_sample(T_(logits, dev), seed, offset, row0, want_probs=True, want_uniforms=True)
    oact, oprobs, ouni = oracle.policy_sample(logits, seed, offset, row0)
    # uniforms: integer Philox -> bit-exact
    np.testing.assert_array_equal(uni.cpu().numpy(), ouni)
    # probabilities: float32 softmax, expf may differ in the last ulp between libm and ocml
    np.testing.assert_allclose(probs.cpu().numpy(), oprobs, rtol=2e-6, atol=1e-9)
    # action indices: bit-exact given the probabilities the device itself used
    np.testing.assert_array_equal(act.cpu().numpy(), oracle.categorical_sample(probs.cpu().numpy(), ouni))
    # sharding invariance: rows [k:] sampled alone with row0+k give the same actions
    act2 = ops.policy_sample(T_(logits[1000:], dev), seed, offset, row0 + 1000)
    np.testing.assert_array_equal(act2.cpu().numpy(), act.cpu().numpy()[1000:])
    # distribution sanity (chi-square-ish): empirical frequencies of a fixed row
    row = np.tile(logits[:1], (200000, 1))
    a = ops.policy_sample(T_(row, dev), 5, 0, 0).cpu().numpy()
    freq = np.bincount(a, minlength=A) / len(a)
    np.testing.assert_allclose(freq, oprobs[0], atol=5e-3)


# ---- fused IMPALA loss (parlhip_impala_loss_f32): loss terms + gradient in one kernel ----
@pytest.mark.parametrize('T,B,A,time_major', [(50, 64, 6, True), (50, 20, 6, False), (7, 3, 4, True),
                                              (130, 5, 18, False), (200, 9, 2, True)])
def test_impala_fused_loss_matches_autograd(dev, T, B, A, time_major):
    """ops.impala_loss == the reference formulas (impala.py:25-79,119-165) evaluated by torch autograd
    in float64 on top of the same V-trace targets: loss sums to 1e-5 relative, gradients w.r.t.
    target logits / values to 1e-5 of their scale."""
    import torch.nn.functional as F
    from parl_amd import ops
    g = torch.Generator(device=dev).manual_seed(T * 1000 + B)
    shp = (T, B) if time_major else (B, T)
    bl = torch.randn(shp + (A, ), device=dev, generator=g)
    tl = (bl + 0.3 * torch.randn(shp + (A, ), device=dev, generator=g)).requires_grad_(True)
    act = torch.randint(0, A, shp, device=dev, generator=g)
    rew = torch.randint(-1, 2, shp, device=dev, generator=g).float()
    dones = torch.rand(shp, device=dev, generator=g) < 0.05
    val = torch.randn(shp, device=dev, generator=g).requires_grad_(True)
    gamma, crho, cpg, vf_c, ent_c = 0.99, 1.0, 1.0, 0.5, -0.01
    out = ops.impala_loss(bl, tl.detach(), act, rew, dones, val.detach(), gamma, crho, cpg, vf_c, ent_c,
                          time_major=time_major)
    assert out is not None
    vs, pg, glog, gval, sums = out
    vs2, pg2 = ops.vtrace_from_logits(bl, tl.detach(), act, rew, dones, val.detach(), gamma, crho, cpg,
                                      time_major=time_major)
    np.testing.assert_allclose(vs.cpu().numpy(), vs2.cpu().numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(pg.cpu().numpy(), pg2.cpu().numpy(), rtol=1e-6, atol=1e-6)
    cut = (lambda x: x[:-1]) if time_major else (lambda x: x[:, :-1])
    tl64, val64 = tl.double(), val.double()
    logp = F.log_softmax(tl64, dim=-1)
    p = logp.exp()
    ent = -(p * logp).sum(-1)
    tlp = logp.gather(-1, act.unsqueeze(-1)).squeeze(-1)
    pi_loss = -(cut(tlp) * pg2.double()).sum()
    vf_loss = 0.5 * ((cut(val64) - vs2.double())**2).sum()
    entropy = cut(ent).sum()
    total = pi_loss + vf_c * vf_loss + ent_c * entropy
    total.backward()
    kl = (p * (logp - F.log_softmax(bl.double(), dim=-1))).sum()
    ref = np.array([float(pi_loss), float(vf_loss), float(entropy), float(kl)])
    np.testing.assert_allclose(sums.cpu().numpy(), ref, rtol=1e-5, atol=1e-4)
    for mine, want in ((glog, tl.grad), (gval, val.grad)):
        scale = float(want.abs().max()) + 1e-12
        assert float((mine.double() - want.double()).abs().max()) <= 1e-5 * scale + 1e-7
    # unsupported shapes are reported, not silently mis-computed
    assert ops.impala_loss(torch.randn((4, 2, 5), device=dev), torch.randn((4, 2, 5), device=dev),
                           torch.zeros((4, 2), dtype=torch.int64, device=dev), torch.zeros((4, 2), device=dev),
                           torch.zeros((4, 2), dtype=torch.bool, device=dev), torch.zeros((4, 2), device=dev),
                           0.99) is None


def test_impala_learn_fused_equals_unfused(dev):
    """IMPALA.learn with the one-kernel loss == the autograd graph of the reference formulas:
    same losses, same parameters after the update (both layouts)."""
    import copy
    import parl_amd as parl
    from parl_amd.models import AtariModel42
    torch.manual_seed(1)
    T, B, A = 10, 6, 6
    for time_major in (True, False):
        m1 = AtariModel42(A).to(dev)
        m2 = copy.deepcopy(m1)
        algs = []
        for m, fused in ((m1, True), (m2, False)):
            alg = parl.algorithms.IMPALA(m, sample_batch_steps=T, gamma=0.99, vf_loss_coeff=0.5,
                                         clip_rho_threshold=1.0, clip_pg_rho_threshold=1.0)
            alg.fused_loss = fused
            algs.append(alg)
        obs = torch.randint(0, 256, (T * B, 4, 42, 42), dtype=torch.uint8, device=dev)
        act = torch.randint(0, A, (T * B, ), device=dev)
        bl = torch.randn((T * B, A), device=dev)
        rew = torch.randint(-1, 2, (T * B, ), device=dev).float()
        dones = torch.rand(T * B, device=dev) < 0.1
        res = [alg.learn(obs, act, bl, rew, dones, 1e-3, -0.01, time_major=time_major) for alg in algs]
        (l1, k1), (l2, k2) = res
        for name in ('total_loss', 'pi_loss', 'vf_loss', 'entropy'):
            np.testing.assert_allclose(float(getattr(l1, name)), float(getattr(l2, name)), rtol=2e-5, atol=1e-3)
        np.testing.assert_allclose(float(k1), float(k2), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(l1.vtrace_returns.vs.cpu().numpy(), l2.vtrace_returns.vs.cpu().numpy(),
                                   rtol=1e-6, atol=1e-6)
        for p1, p2 in zip(m1.parameters(), m2.parameters()):
            np.testing.assert_allclose(p1.detach().cpu().numpy(), p2.detach().cpu().numpy(), rtol=1e-3, atol=2e-5)


def test_impala_learn_in_row_chunks_equals_one_pass(dev):
    """IMPALA.max_learn_rows: the update accumulated over chunks of whole sequences == the one-pass
    update (losses are sums, impala.py:67-79), both layouts."""
    import copy
    import parl_amd as parl
    from parl_amd.models import AtariModel42
    torch.manual_seed(0)
    T, B, A = 10, 12, 6
    base = AtariModel42(A).to(dev)
    obs = torch.randint(0, 256, (T * B, 4, 42, 42), dtype=torch.uint8, device=dev)
    act = torch.randint(0, A, (T * B, ), device=dev)
    bl = torch.randn((T * B, A), device=dev)
    rew = torch.randn(T * B, device=dev)
    dn = torch.rand(T * B, device=dev) < 0.05
    for tm in (True, False):
        outs = []
        for rows, mode in ((None, 'forward'), (3 * T, 'forward'), (5 * T, 'accumulate')):
            m = copy.deepcopy(base)
            alg = parl.algorithms.IMPALA(m, sample_batch_steps=T, gamma=0.99, vf_loss_coeff=0.5,
                                         clip_rho_threshold=1.0, clip_pg_rho_threshold=1.0)
            alg.max_learn_rows, alg.learn_chunk_mode = rows, mode
            loss, kl = alg.learn(obs, act, bl, rew, dn, 1e-3, -0.01, time_major=tm)
            # the (clipped) gradients the Adam step consumed; Adam's first step is lr * sign-like, so the
            # parameters themselves amplify rounding differences of tiny gradients
            outs.append((float(loss.total_loss.detach()), [p.grad.detach().clone() for p in m.parameters()]))
        for other in outs[1:]:
            assert abs(outs[0][0] - other[0]) <= 1e-4 * abs(outs[0][0])
            for a, b in zip(outs[0][1], other[1]):
                assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max()) + 1e-7


@pytest.mark.parametrize('T,B,A', [(50, 1024, 6), (50, 37, 4), (10, 5, 6), (64, 9, 4), (2, 3, 6)])
def test_impala_heads_loss_matches_heads_plus_fused_loss(dev, T, B, A):
    """parlhip_impala_heads_loss_f32 (policy_fc + value_fc + loss + the heads' backward, one kernel) against
    its two-kernel twin: torch float32 heads (rocBLAS) + ops.impala_loss behind them + torch autograd through
    the heads.  A consistency check between two HIP paths (hence the looser tolerance: two float32 head
    GEMMs with different summation orders); the check against the CPU oracle / float64 is
    test_impala_heads_loss_vs_oracle_and_f64_autograd below."""
    from parl_amd import ops
    torch.manual_seed(T * 1000 + B)
    H = 256
    hd = torch.relu(torch.randn(T, B, H, device=dev))
    wp, bp = torch.randn(A, H, device=dev) * 0.1, torch.randn(A, device=dev) * 0.1
    wv, bv = torch.randn(1, H, device=dev) * 0.05, torch.randn(1, device=dev) * 0.1
    bl = torch.randn(T, B, A, device=dev)
    act = torch.randint(0, A, (T, B), device=dev)
    rew = torch.randn(T, B, device=dev)
    dn = torch.rand(T, B, device=dev) < 0.05
    out = ops.impala_heads_loss(hd, wp, bp, wv, bv, bl, act, rew, dn, 0.99, 1.0, 1.0, 0.5, -0.01)
    assert out is not None
    vs, pg, gh, gwp, gbp, gwv, gbv, sums = out
    # reference: the same heads by torch (float32 GEMMs), the fused loss kernel behind them
    hd_r = hd.clone().requires_grad_(True)
    prm = [x.clone().requires_grad_(True) for x in (wp, bp, wv, bv)]
    logits = torch.nn.functional.linear(hd_r, prm[0], prm[1])
    values = torch.nn.functional.linear(hd_r, prm[2], prm[3]).squeeze(-1)
    rvs, rpg, glog, gval, rsums = ops.impala_loss(bl, logits.detach(), act, rew, dn, values.detach(), 0.99, 1.0, 1.0,
                                                  0.5, -0.01, time_major=True)
    torch.autograd.backward([logits, values], [glog, gval])
    np.testing.assert_allclose(sums.cpu().numpy(), rsums.cpu().numpy(), rtol=2e-5, atol=1e-3)
    np.testing.assert_allclose(vs.cpu().numpy(), rvs.cpu().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(pg.cpu().numpy(), rpg.cpu().numpy(), rtol=1e-4, atol=1e-4)

    def close(a, b, name):
        a, b = a.double().cpu().numpy(), b.double().cpu().numpy()
        scale = np.abs(b).max() + 1e-12
        assert np.abs(a - b).max() <= 2e-4 * scale, (name, np.abs(a - b).max(), scale)

    close(gh, hd_r.grad, 'grad_hidden')
    close(gwp, prm[0].grad, 'grad_w_policy')
    close(gbp, prm[1].grad, 'grad_b_policy')
    close(gwv.reshape(1, H), prm[2].grad, 'grad_w_value')
    close(gbv, prm[3].grad, 'grad_b_value')
    # deterministic apart from the f64 atomics of the four sums
    out2 = ops.impala_heads_loss(hd, wp, bp, wv, bv, bl, act, rew, dn, 0.99, 1.0, 1.0, 0.5, -0.01)
    for a, b in zip(out[:7], out2[:7]):
        assert torch.equal(a, b)


@pytest.mark.parametrize('T,B,A', [(50, 1024, 6), (50, 1024, 4), (50, 37, 6), (7, 5, 4)])
def test_impala_heads_loss_vs_oracle_and_f64_autograd(dev, T, B, A):
    """The kernel the bench's learner runs, checked directly against the CPU path at north_star's 1e-5:
      * heads (policy_fc / value_fc, atari_model.py:44-57,73-90) evaluated on the HOST in float64 from the same
        h, W, b, rounded to float32 (what a float32 framework hands to the loss);
      * V-trace targets from the C oracle (vtrace.py:99-137 + impala.py:59,167-194) on those logits / values:
        vs / pg_adv rtol = atol = 1e-5;
      * the loss sums and KL (impala.py:59-79, 151-162) in float64 on the host: 1e-5 relative;
      * d total / d h, d W, d b by float64 autograd through float64 heads with the oracle's targets as constants
        (vtrace.py:36 @no_grad): 1e-5 of each gradient's scale."""
    import torch.nn.functional as F
    from parl_amd import ops
    from oracle import c_oracle
    torch.manual_seed(T * 1000 + B + A)
    H = 256
    hd = torch.relu(torch.randn(T, B, H))
    wp, bp = torch.randn(A, H) * 0.1, torch.randn(A) * 0.1
    wv, bv = torch.randn(1, H) * 0.05, torch.randn(1) * 0.1
    bl = torch.randn(T, B, A)
    act = torch.randint(0, A, (T, B))
    rew = torch.randint(-1, 2, (T, B)).float()
    dn = torch.rand(T, B) < 0.05
    gamma, crho, cpg, vf_c, ent_c = 0.99, 1.0, 1.0, 0.5, -0.01
    d = lambda x: x.to(dev)
    out = ops.impala_heads_loss(d(hd), d(wp), d(bp), d(wv), d(bv), d(bl), d(act), d(rew), d(dn), gamma, crho, cpg,
                                vf_c, ent_c)
    assert out is not None
    vs, pg, gh, gwp, gbp, gwv, gbv, sums = [x.cpu() for x in out]
    # host, float64
    h64 = hd.double().requires_grad_(True)
    prm = [x.double().requires_grad_(True) for x in (wp, bp, wv, bv)]
    logits64 = F.linear(h64, prm[0], prm[1])
    values64 = F.linear(h64, prm[2], prm[3]).squeeze(-1)
    ovs, opg, _, _ = c_oracle.vtrace_from_logits(bl.numpy(), logits64.detach().float().numpy(), act.numpy(),
                                                 rew.numpy(), dn.numpy(), values64.detach().float().numpy(), gamma,
                                                 crho, cpg)
    np.testing.assert_allclose(vs.numpy(), ovs, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(pg.numpy(), opg, rtol=1e-5, atol=1e-5)
    logp = F.log_softmax(logits64, dim=-1)
    p = logp.exp()
    ent = -(p * logp).sum(-1)
    tlp = logp.gather(-1, act.unsqueeze(-1)).squeeze(-1)
    pi_loss = -(tlp[:-1] * torch.from_numpy(opg).double()).sum()
    vf_loss = 0.5 * ((values64[:-1] - torch.from_numpy(ovs).double())**2).sum()
    entropy = ent[:-1].sum()
    kl = (p * (logp - F.log_softmax(bl.double(), dim=-1))).sum()
    (pi_loss + vf_c * vf_loss + ent_c * entropy).backward()
    ref = np.array([float(pi_loss), float(vf_loss), float(entropy), float(kl)])
    np.testing.assert_allclose(sums.numpy(), ref, rtol=1e-5, atol=1e-5 * float(np.abs(ref).max()))
    for mine, want, name in ((gh, h64.grad, 'd h'), (gwp, prm[0].grad, 'd W_policy'), (gbp, prm[1].grad, 'd b_policy'),
                             (gwv.reshape(1, H), prm[2].grad, 'd W_value'), (gbv, prm[3].grad, 'd b_value')):
        scale = float(want.abs().max()) + 1e-30
        err = float((mine.double() - want).abs().max())
        assert err <= 1e-5 * scale, (name, err, scale)


def test_impala_learn_fused_heads_equals_framework_heads(dev):
    import copy
    import parl_amd as parl
    from parl_amd.models import AtariModel42
    torch.manual_seed(1)
    T, B, A = 50, 24, 6
    base = AtariModel42(A).to(dev)
    with torch.no_grad():  # N(0,1) heads (atari_model.py:44-57) make huge logits on a random trunk: tame them
        base.policy_fc.weight.mul_(0.05)
        base.value_fc.weight.mul_(0.05)
    obs = torch.randint(0, 256, (T * B, 4, 42, 42), dtype=torch.uint8, device=dev)
    act = torch.randint(0, A, (T * B, ), device=dev)
    bl = torch.randn((T * B, A), device=dev)
    rew = torch.randn(T * B, device=dev)
    dn = torch.rand(T * B, device=dev) < 0.05
    outs = []
    for fused_heads, rows in ((False, None), (True, None), (True, 8 * T)):
        m = copy.deepcopy(base)
        alg = parl.algorithms.IMPALA(m, sample_batch_steps=T, gamma=0.99, vf_loss_coeff=0.5,
                                     clip_rho_threshold=1.0, clip_pg_rho_threshold=1.0)
        alg.fused_heads, alg.max_learn_rows = fused_heads, rows
        loss, kl = alg.learn(obs, act, bl, rew, dn, 1e-3, -0.01, time_major=True)
        outs.append((float(loss.total_loss.detach()), float(kl), [p.grad.detach().clone() for p in m.parameters()]))
    for other in outs[1:]:
        assert abs(outs[0][0] - other[0]) <= 1e-4 * abs(outs[0][0])
        assert abs(outs[0][1] - other[1]) <= 1e-4 * abs(outs[0][1]) + 1e-7
        for a, b in zip(outs[0][2], other[2]):
            assert float((a - b).abs().max()) <= 2e-4 * float(a.abs().max()) + 1e-7


@pytest.mark.parametrize('B,A', [(1024, 6), (37, 4), (5, 18)])
def test_policy_head_sample_equals_head_then_policy_sample(dev, B, A):
    """parlhip_policy_head_sample_f32 (the actors' policy_fc + draw in one launch): logits against a float64 head
    (1e-5 of scale), and the actions bit-exact against parlhip_policy_sample_f32 on the logits it wrote (same
    Philox stream, same softmax / inverse-CDF arithmetic) — and through it against numpy's choice."""
    from parl_amd import ops
    torch.manual_seed(B + A)
    h = torch.relu(torch.randn(B, 256, device=dev))
    w, b = torch.randn(A, 256, device=dev) * 0.1, torch.randn(A, device=dev)
    logits = torch.zeros(B, A, device=dev)
    act = torch.zeros(B, dtype=torch.int64, device=dev)
    assert ops.policy_head_sample_into(h, w, b, logits, act, seed=7, offset=3, row0=11)
    ref = (h.double() @ w.double().t() + b.double()).cpu().numpy()
    got = logits.cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()
    act2 = ops.policy_sample(logits, 7, 3, row0=11)
    assert torch.equal(act, act2)
    assert int(act.min()) >= 0 and int(act.max()) < A
    # 512 hidden units: no instantiation, the caller falls back
    assert not ops.policy_head_sample_into(torch.zeros(4, 512, device=dev), torch.zeros(A, 512, device=dev), b,
                                           torch.zeros(4, A, device=dev), torch.zeros(4, dtype=torch.int64, device=dev),
                                           1, 2)
