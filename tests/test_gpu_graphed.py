"""IMPALA.learn at the reference's learner batch (train_batch_size = 1000 rows, impala_config.py:31) as a
hipGraph replay (parl_amd.algorithms.impala.graphed.GraphedLearn): the same update as the eager call."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _batch(dev, T, B, A, dim, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    N = T * B
    return {'obs': torch.randint(0, 256, (N, 4, dim, dim), dtype=torch.uint8, device=dev, generator=g),
            'actions': torch.randint(0, A, (N, ), device=dev, generator=g),
            'behaviour_logits': torch.randn((N, A), device=dev, generator=g),
            'rewards': torch.randint(-1, 2, (N, ), device=dev, generator=g).float(),
            'dones': torch.rand(N, device=dev, generator=g) < 0.02}


@pytest.mark.parametrize('dim,B', [(42, 20), (42, 24), (84, 4)])
def test_graphed_learn_equals_eager_learn(dev, dim, B):
    """three updates with a changing learning rate (the piecewise schedule is a device scalar, no re-capture) on
    three different batches: same losses, same parameters as IMPALA.learn(time_major=True) issued eagerly; the
    warm-up iterations of the capture leave parameters and optimizer state untouched"""
    import parl_amd as parl
    from parl_amd.algorithms.impala.graphed import GraphedLearn
    from parl_amd.models import AtariModel42, AtariModel84
    torch.manual_seed(3)
    T, A = 50, 6
    base = (AtariModel42 if dim == 42 else AtariModel84)(A).to(dev)
    with torch.no_grad():
        base.policy_fc.weight.mul_(0.05)
        base.value_fc.weight.mul_(0.05)
    mk = lambda m: parl.algorithms.IMPALA(m, sample_batch_steps=T, gamma=0.99, vf_loss_coeff=0.5,  # noqa: E731
                                          clip_rho_threshold=1.0, clip_pg_rho_threshold=1.0)
    m_e, m_g = copy.deepcopy(base), copy.deepcopy(base)
    alg_e, alg_g = mk(m_e), mk(m_g)
    gl = GraphedLearn(alg_g, B, (4, dim, dim), A, entropy_coeff=-0.01)
    for a, b in zip(m_g.parameters(), base.parameters()):
        assert torch.equal(a, b), 'capture changed the parameters'
    E = B + 3  # the graph takes a [T, b0:b0+B] slice of a wider rollout
    for i, lr in enumerate((1e-3, 1e-3, 5e-4)):
        wide = _batch(dev, T, E, A, dim, 10 + i)
        b0 = i
        cut = lambda x: x.view((T, E) + tuple(x.shape[1:]))[:, b0:b0 + B].reshape((T * B, ) + tuple(x.shape[1:]))  # noqa: E731
        loss, kl = alg_e.learn(cut(wide['obs']), cut(wide['actions']), cut(wide['behaviour_logits']),
                               cut(wide['rewards']), cut(wide['dones']), lr, -0.01, time_major=True)
        gl.load(wide, b0, E)
        gl.replay(lr, -0.01)
        want = np.array([float(loss.total_loss), float(loss.pi_loss), float(loss.vf_loss), float(loss.entropy),
                         float(kl)])
        got = gl.out.cpu().numpy()
        # pi_loss is a sum of 1000 signed terms that nearly cancels: absolute tolerance on the terms' scale
        np.testing.assert_allclose(got[:5], want, rtol=1e-5, atol=1e-5 * float(np.abs(want).max()))
        assert got[5] == 1.0
        for (n, a), b in zip(m_g.named_parameters(), m_e.parameters()):
            # same kernels, same gradients; fused capturable Adam vs the foreach implementation differ by rounding,
            # which Adam's normalised step turns into a small fraction of lr = 1e-3 per update (measured: 2.5e-6)
            assert float((a - b).abs().max()) <= 2e-5 + 1e-5 * float(b.abs().max()), (i, n)
    means, n = gl.pop_stats()
    assert n == 3 and len(means) == 5 and np.isfinite(means).all()
    assert gl.pop_stats()[1] == 0


def test_async_pipeline_with_reference_train_batch(dev):
    """AsyncActorLearner(train_batch_size=...): a 16-env rollout consumed as 16 // 3 = 5 updates (3, 3, 3, 3, 4
    sequences), schedulers stepped once per update; parameters equal the same sub-batches learned eagerly"""
    import parl_amd as parl
    from parl_amd.env import DeviceVectorEnv
    from parl_amd.models import AtariModel42
    from parl_amd.rollout import AsyncActorLearner
    from parl_amd.utils import PiecewiseScheduler
    torch.manual_seed(5)
    T, E = 8, 16
    env = DeviceVectorEnv('PongNoFrameskip-v4', E, dim=42, horizon=T, seed=3, device=dev)
    model = AtariModel42(env.act_dim).to(dev)
    with torch.no_grad():
        model.policy_fc.weight.mul_(0.05)
        model.value_fc.weight.mul_(0.05)
    twin = copy.deepcopy(model)
    mk = lambda m: parl.algorithms.IMPALA(m, sample_batch_steps=T, gamma=0.99, vf_loss_coeff=0.5,  # noqa: E731
                                          clip_rho_threshold=1.0, clip_pg_rho_threshold=1.0)
    alg, alg_t = mk(model), mk(twin)
    pipe = AsyncActorLearner(alg, [env], T, seed=9, train_batch_size=3 * T)
    assert pipe.sub_batches == [(0, 3), (3, 3), (6, 3), (9, 3), (12, 4)]
    lr_s = PiecewiseScheduler([(0, 1e-3), (3, 5e-4)])
    pipe.prime()
    pipe.actor_stream.synchronize()
    batch = {k: v.clone() for k, v in pipe.pending[0][0].items()}
    torch.cuda.synchronize()
    loss, kl = pipe.step(lr_s, -0.01)
    pipe.synchronize()
    assert pipe.updates == 5 and np.isfinite(float(loss.total_loss))
    lr_t = PiecewiseScheduler([(0, 1e-3), (3, 5e-4)])
    for b0, nb in pipe.sub_batches:
        cut = lambda x: x.view((T, E) + tuple(x.shape[1:]))[:, b0:b0 + nb].reshape((T * nb, ) + tuple(x.shape[1:]))  # noqa: E731
        alg_t.learn(cut(batch['obs']), cut(batch['actions']), cut(batch['behaviour_logits']), cut(batch['rewards']),
                    cut(batch['dones']), lr_t.step(), -0.01, time_major=True)
    for (n, a), b in zip(model.named_parameters(), twin.parameters()):
        assert float((a - b).abs().max()) <= 5e-5 + 2e-5 * float(b.abs().max()), n
    means, n = pipe.pop_learn_stats()
    assert n == 5 and np.isfinite(means).all()
    env.check_faults()


def test_graphed_pipeline_checkpoint_resumes_bit_identically(dev):
    """train_batch_size mode: the optimizer state lives at addresses the captured graphs hold, so a resume copies
    the checkpoint INTO it (graphed.load_optimizer_state_inplace) — save after two steps, two more steps, restore,
    the same two steps again: every parameter bit-identical (envs, samplers, pending batch, actor snapshot from
    AsyncActorLearner.state_dict; mid-rollout weight refresh active)."""
    import parl_amd as parl
    from parl_amd.algorithms.impala.graphed import load_optimizer_state_inplace
    from parl_amd.env import DeviceVectorEnv
    from parl_amd.models import AtariModel42
    from parl_amd.rollout import AsyncActorLearner
    from parl_amd.utils import PiecewiseScheduler
    torch.manual_seed(2)
    T, E = 10, 12
    env = DeviceVectorEnv('BreakoutNoFrameskip-v4', E, dim=42, horizon=T, seed=13, device=dev)
    model = AtariModel42(env.act_dim).to(dev)
    with torch.no_grad():
        model.policy_fc.weight.mul_(0.05)
        model.value_fc.weight.mul_(0.05)
    alg = parl.algorithms.IMPALA(model, sample_batch_steps=T, gamma=0.99, vf_loss_coeff=0.5,
                                 clip_rho_threshold=1.0, clip_pg_rho_threshold=1.0)
    # explicit refresh points: deterministic (the default 'auto' calibrates them from measured timings)
    pipe = AsyncActorLearner(alg, [env], T, seed=3, train_batch_size=4 * T, refresh_points=[(2, 1), (4, 3)])
    assert pipe.refresh_points == [(2, 1), (4, 3)] and len(pipe.sub_batches) == 3
    lr_s = PiecewiseScheduler([(0, 1e-3), (7, 5e-4)])
    for _ in range(2):
        pipe.step(lr_s, -0.01)
    sd_pipe = pipe.state_dict()
    sd_model = {k: v.detach().clone() for k, v in model.state_dict().items()}
    osd = alg.optimizer.state_dict()
    sd_opt = {'state': {i: {k: (v.detach().clone() if isinstance(v, torch.Tensor) else v) for k, v in st.items()}
                        for i, st in osd['state'].items()},
              'param_groups': [{k: (float(v) if isinstance(v, torch.Tensor) else v) for k, v in g.items()}
                               for g in osd['param_groups']]}
    lr_state = (lr_s.cur_index, lr_s.cur_step, lr_s.cur_value)

    def run():
        for _ in range(2):
            pipe.step(lr_s, -0.01)
        pipe.synchronize()
        return [p.detach().clone() for p in model.parameters()]

    p1 = run()
    assert any(not torch.equal(a, b) for a, b in zip(p1, sd_model.values()))
    model.load_state_dict(sd_model)
    load_optimizer_state_inplace(alg.optimizer, sd_opt)
    pipe.load_state_dict(sd_pipe)
    lr_s.cur_index, lr_s.cur_step, lr_s.cur_value = lr_state
    p2 = run()
    for a, b in zip(p1, p2):
        assert torch.equal(a, b)
    env.check_faults()


def test_refresh_points_are_calibrated_after_the_second_step(dev):
    """refresh_points='auto': step 2 is timed, step 3 fixes the points — at most three, at env steps
    T/5, 2T/5, 3T/5, each asking for no more updates than a pass has, increasing; the pipeline keeps running and
    the calibrated points travel in its state_dict"""
    import parl_amd as parl
    from parl_amd.env import DeviceVectorEnv
    from parl_amd.models import AtariModel42
    from parl_amd.rollout import AsyncActorLearner
    torch.manual_seed(4)
    T, E = 20, 24
    env = DeviceVectorEnv('PongNoFrameskip-v4', E, dim=42, horizon=T, seed=2, device=dev)
    model = AtariModel42(env.act_dim).to(dev)
    with torch.no_grad():
        model.policy_fc.weight.mul_(0.05)
        model.value_fc.weight.mul_(0.05)
    alg = parl.algorithms.IMPALA(model, sample_batch_steps=T, gamma=0.99, vf_loss_coeff=0.5,
                                 clip_rho_threshold=1.0, clip_pg_rho_threshold=1.0)
    pipe = AsyncActorLearner(alg, [env], T, seed=1, train_batch_size=2 * T, refresh_points='auto')
    n = len(pipe.sub_batches)
    assert n == 12 and pipe.refresh_points == []
    for i in range(5):
        loss, kl = pipe.step(1e-3, -0.01)
        if i < 2:
            assert pipe.refresh_points == []  # not before the timed step has been read
    pipe.synchronize()
    pts = pipe.refresh_points
    assert len(pts) <= 3 and all(s in (T // 5, 2 * T // 5, 3 * T // 5) and 2 <= u <= n for s, u in pts)
    assert [u for _, u in pts] == sorted({u for _, u in pts})
    assert pipe.refresh_calibration['rollout_ms'] > 0 and pipe.refresh_calibration['learner_pass_ms'] > 0
    assert np.isfinite(float(loss.total_loss)) and pipe.updates == 5 * n
    assert [tuple(x) for x in pipe.state_dict()['refresh_points']] == pts
    env.check_faults()


def test_default_refresh_points_are_a_function_of_the_shapes_only(dev):
    """the default ('fixed'): the points depend on T, the updates per rollout and the frame size — not on anything
    timed — so the same command line is the same run on every box; a checkpoint whose points no pass can honour
    (an update count beyond the pass, a step outside the rollout) is refused instead of silently never waited for"""
    import parl_amd as parl
    from parl_amd.env import DeviceVectorEnv
    from parl_amd.models import AtariModel42
    from parl_amd.rollout import AsyncActorLearner, fixed_refresh_points
    assert fixed_refresh_points(50, 51, 42) == [(10, 14), (20, 28), (30, 42)]
    assert fixed_refresh_points(50, 51, 84) == [(10, 4), (20, 9), (30, 13)]
    torch.manual_seed(4)
    T, E = 20, 24
    env = DeviceVectorEnv('PongNoFrameskip-v4', E, dim=42, horizon=T, seed=2, device=dev)
    model = AtariModel42(env.act_dim).to(dev)
    alg = parl.algorithms.IMPALA(model, sample_batch_steps=T, gamma=0.99, vf_loss_coeff=0.5,
                                 clip_rho_threshold=1.0, clip_pg_rho_threshold=1.0)
    pipe = AsyncActorLearner(alg, [env], T, seed=1, train_batch_size=2 * T)
    assert pipe.refresh_points == fixed_refresh_points(T, 12, 42) == [(4, 3), (8, 6), (12, 10)]
    for _ in range(3):
        loss, kl = pipe.step(1e-3, -0.01)
    pipe.synchronize()
    assert np.isfinite(float(loss.total_loss)) and pipe.updates == 36
    sd = pipe.state_dict()
    for bad in ([[4, 13]], [[20, 3]], [[0, 3]], [[4, 3], [4, 6]]):
        with pytest.raises(ValueError):
            pipe.load_state_dict(dict(sd, refresh_points=bad))
    pipe.load_state_dict(sd)
    assert pipe.refresh_points == [(4, 3), (8, 6), (12, 10)]
    with pytest.raises(RuntimeError, match='load_optimizer_state_inplace'):
        alg.optimizer.load_state_dict(alg.optimizer.state_dict())  # would detach the graphs from the state
    env.check_faults()


def test_rollout_segments_as_hipgraphs_equal_the_eager_steps(dev):
    """DeviceRollout.collect_segment(graph=True): a segment of env steps replayed as ONE hipGraph (the Philox
    offset of the rollout's first step read from device memory) produces bit for bit the batches of the eager
    steps — across the eager first run of every (buffer, segment), the captures and several replays, with the
    actor weights rewritten between rollouts (the graph refers to the parameters by address) and a weight
    refresh between two segments as AsyncActorLearner does it."""
    from parl_amd.env import DeviceVectorEnv
    from parl_amd.models import AtariModel42
    from parl_amd.rollout import DeviceRollout
    E, T = 48, 12
    torch.manual_seed(5)
    base = AtariModel42(4).to(dev)
    with torch.no_grad():
        base.policy_fc.weight.mul_(0.05)
    outs = []
    for graph in (False, True):
        env = DeviceVectorEnv('BreakoutNoFrameskip-v4', E, dim=42, horizon=T, seed=11, device=dev, max_episode_steps=60)
        ro = DeviceRollout(env, T, seed=7, n_buffers=2)
        m = copy.deepcopy(base)
        for p in m.parameters():
            p.requires_grad_(False)
        st = torch.cuda.Stream(device=dev)
        st.wait_stream(torch.cuda.current_stream(dev))
        got = []
        with torch.cuda.stream(st):
            for it in range(7):
                ro.collect_begin()
                ro.collect_segment(m, 0, 5, graph=graph)
                with torch.no_grad():   # a mid-rollout refresh: in-place, same addresses
                    m.policy_fc.bias.add_(0.01 * (it + 1))
                ro.collect_segment(m, 5, T, graph=graph)
                b = ro.collect_end()
                got.append({k: v.clone() for k, v in b.items()})
                with torch.no_grad():
                    m.policy_fc.weight.mul_(1.01)
        st.synchronize()
        env.check_faults()
        if graph:
            assert len(ro._graphs) == 4, ro._graphs.keys()   # 2 buffers x 2 segments were captured
        outs.append((got, ro.step_count, ro.pop_episode_stats()))
    (a, sa, ea), (b, sb, eb) = outs
    assert sa == sb == 7 * T and ea == eb and ea[0] > 0
    for it, (x, y) in enumerate(zip(a, b)):
        for k in x:
            assert torch.equal(x[k], y[k]), (it, k)


def test_pipeline_with_observations_left_in_the_frame_rings_equals_materialised_batches(dev, monkeypatch):
    """AsyncActorLearner keeps a rollout's observations in the env's frame rings (one ring per trajectory buffer,
    rollout.RingBatch) and every 1000-row-style update gathers its sequences straight from there — against the
    pipeline that materialises a [T*E, 4, d, d] batch per rollout (PARL_AMD_LAZY_OBS=0): the same parameters, bit
    for bit, after several rollouts of graph-replayed updates with mid-rollout weight refreshes, and in the
    one-update-per-rollout mode."""
    import parl_amd as parl
    from parl_amd.env import DeviceVectorEnv
    from parl_amd.models import AtariModel42
    from parl_amd.rollout import AsyncActorLearner, RingBatch
    E, T = 24, 10
    for tb in (4 * T, None):
        outs = []
        for lazy in ('1', '0'):
            monkeypatch.setenv('PARL_AMD_LAZY_OBS', lazy)
            torch.manual_seed(0)
            env = DeviceVectorEnv('BreakoutNoFrameskip-v4', E, dim=42, horizon=T, seed=13, device=dev, max_episode_steps=50)
            model = AtariModel42(env.act_dim).to(dev)
            with torch.no_grad():
                model.policy_fc.weight.mul_(0.05)
                model.value_fc.weight.mul_(0.05)
            alg = parl.algorithms.IMPALA(model, sample_batch_steps=T, gamma=0.99, vf_loss_coeff=0.5,
                                         clip_rho_threshold=1.0, clip_pg_rho_threshold=1.0)
            pipe = AsyncActorLearner(alg, [env], T, seed=3, train_batch_size=tb)
            for _ in range(5):
                pipe.step(1e-3, -0.01)
            pipe.synchronize()
            obs = pipe.pending[0][0]['obs']
            assert isinstance(obs, RingBatch) == (lazy == '1')
            if lazy == '1':
                assert len(env._rings) == 2 and 'obs' not in pipe.rollout._bufs[0]
            outs.append(([p.detach().clone() for p in model.parameters()], obs.clone(), pipe.updates))
            env.check_faults()
        (pa, oa, ua), (pb, ob, ub) = outs
        assert ua == ub and torch.equal(oa, ob)
        for a, b in zip(pa, pb):
            assert torch.equal(a, b)
