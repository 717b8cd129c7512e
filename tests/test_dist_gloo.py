"""N>1 path on CPU: world_size-2 gloo processes.  Covers the host-side exchange
logic (flat gradient all-reduce with SUM semantics, clip on the REDUCED gradient, identical Adam
step on every rank, small-tensor all-gather, max-over-ranks timing) and the env sharding contract
(env ids rank*E .. rank*E+E-1 keep their RNG streams, checked on the CPU oracle env)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import torch.nn as nn
    import parl_amd as parl
    from parl_amd import dist as pdist

    r, l, w = pdist.init(backend='gloo')
    assert (r, w) == (rank, world)

    class M(parl.Model):
        def __init__(self):
            super(M, self).__init__()
            self.fc = nn.Linear(4, 8)
            self.pi = nn.Linear(8, 3)
            self.v = nn.Linear(8, 1)

        def policy(self, o):
            return self.pi(torch.tanh(self.fc(o)))

        def value(self, o):
            return self.v(torch.tanh(self.fc(o))).squeeze(1)

        def policy_and_value(self, o):
            h = torch.tanh(self.fc(o))
            return self.pi(h), self.v(h).squeeze(1)

    torch.manual_seed(100 + rank)  # different init per rank: broadcast must fix it
    m = M()
    pdist.broadcast_model(m)
    alg = parl.algorithms.A2C(m, vf_loss_coeff=0.5)
    alg.grad_hook = pdist.FlatGradAllReduce(m)
    g = torch.Generator().manual_seed(7)
    obs = torch.randn(2 * 6, 4, generator=g)
    act = torch.randint(0, 3, (2 * 6, ), generator=g)
    adv, tgt = torch.randn(2 * 6, generator=g), torch.randn(2 * 6, generator=g)
    sl = slice(rank * 6, rank * 6 + 6)  # this rank's shard of the union batch
    for _ in range(3):
        alg.learn(obs[sl], act[sl], adv[sl] * 30, tgt[sl], 1e-2, -0.01)  # *30: make the clip bite
    gathered = pdist.all_gather_small({'x': torch.full((2, 3), float(rank))})
    tmax = pdist.all_reduce_max_scalar(1.0 + rank)
    pdist.barrier()
    q.put((rank, {k: v.numpy().copy() for k, v in m.state_dict().items()}, gathered['x'].numpy().copy(), tmax))


def test_data_parallel_matches_single_learner_on_union_batch(tmp_path):
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sd0, sd1 = res[0][1], res[1][1]
    for k in sd0:  # identical parameters on every rank after DP updates
        assert np.array_equal(sd0[k], sd1[k]), k
    assert np.array_equal(res[0][2], np.array([[[0.] * 3] * 2, [[1.] * 3] * 2], np.float32))
    assert res[0][3] == 2.0 and res[1][3] == 2.0

    # single learner on the union batch, starting from rank 0's init: sums of losses => the DP
    # gradient is the SUM over ranks (impala.py:67-79 / a2c.py:67-79), clip applies after the sum
    import torch.nn as nn
    import parl_amd as parl

    class M(parl.Model):
        def __init__(self):
            super(M, self).__init__()
            self.fc = nn.Linear(4, 8)
            self.pi = nn.Linear(8, 3)
            self.v = nn.Linear(8, 1)

        def policy_and_value(self, o):
            h = torch.tanh(self.fc(o))
            return self.pi(h), self.v(h).squeeze(1)

        def policy(self, o):
            return self.policy_and_value(o)[0]

        def value(self, o):
            return self.policy_and_value(o)[1]

    torch.manual_seed(100)
    m = M()
    alg = parl.algorithms.A2C(m, vf_loss_coeff=0.5)
    g = torch.Generator().manual_seed(7)
    obs = torch.randn(12, 4, generator=g)
    act = torch.randint(0, 3, (12, ), generator=g)
    adv, tgt = torch.randn(12, generator=g), torch.randn(12, generator=g)
    for _ in range(3):
        alg.learn(obs, act, adv * 30, tgt, 1e-2, -0.01)
    for k, v in m.state_dict().items():
        np.testing.assert_allclose(sd0[k], v.numpy(), rtol=2e-5, atol=2e-6)


def test_env_sharding_keeps_rng_streams(oracle):
    """rank r owns env ids r*E..r*E+E-1: a shard started at env_id0 reproduces the corresponding
    envs of the unsharded vector env (same noop streams, same trajectories)."""
    from parl_amd.env import find_rom
    try:
        rom = find_rom('breakout')
    except FileNotFoundError:
        pytest.skip('cartridge not present')
    full = oracle.VecEnv(rom, 'breakout', 4, 42, seed=9)
    shard = oracle.VecEnv(rom, 'breakout', 2, 42, seed=9, env_id0=2)
    assert np.array_equal(full.reset()[2:], shard.reset())
    rng = np.random.default_rng(0)
    for _ in range(60):
        a = rng.integers(0, 4, 4)
        o, r, d = full.step(a)
        o2, r2, d2 = shard.step(a[2:])
        assert np.array_equal(o[2:], o2) and np.array_equal(r[2:], r2) and np.array_equal(d[2:], d2)


@pytest.mark.gpu
def test_device_env_sharding_keeps_rng_streams(dev):
    from parl_amd.env import DeviceVectorEnv
    full = DeviceVectorEnv('BreakoutNoFrameskip-v4', 4, dim=42, horizon=8, seed=9, device=dev)
    shard = DeviceVectorEnv('BreakoutNoFrameskip-v4', 2, dim=42, horizon=8, seed=9, env_id0=2, device=dev)
    assert torch.equal(full.reset()[2:], shard.reset())
    g = torch.Generator().manual_seed(0)
    for _ in range(120):
        a = torch.randint(0, 4, (4, ), generator=g).to(dev)
        o, r, d, _ = full.step(a)
        o2, r2, d2, _ = shard.step(a[2:].contiguous())
        assert torch.equal(o[2:], o2) and torch.equal(r[2:], r2) and torch.equal(d[2:], d2)


# ---- PPO: mean-reduced losses => the data-parallel gradient is the MEAN over ranks ----
def _ppo_model():
    import torch.nn as nn
    import parl_amd as parl

    class M(parl.Model):
        def __init__(self):
            super(M, self).__init__()
            self.fc = nn.Linear(4, 8)
            self.pi = nn.Linear(8, 2)
            self.v = nn.Linear(8, 1)
            self.logstd = nn.Parameter(torch.zeros(1, 2))

        def policy(self, o):
            return self.pi(torch.tanh(self.fc(o))), torch.exp(self.logstd)

        def value(self, o):
            return self.v(torch.tanh(self.fc(o)))

    return M()


def _ppo_batch():
    g = torch.Generator().manual_seed(11)
    n = 16
    return dict(obs=torch.randn(n, 4, generator=g), act=torch.randn(n, 2, generator=g),
                val=torch.randn(n, generator=g), ret=torch.randn(n, generator=g),
                logp=torch.randn(n, generator=g) * 0.1 - 2.0, adv=torch.randn(n, generator=g))


def _ppo_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import parl_amd as parl
    from parl_amd import dist as pdist
    pdist.init(backend='gloo')
    torch.manual_seed(200 + rank)
    m = _ppo_model()
    pdist.broadcast_model(m)
    # norm_adv=False: the per-minibatch normalisation is the HIP kernel (no CPU path); the exchange
    # logic under test is independent of it
    alg = parl.algorithms.PPO(m, clip_param=0.2, entropy_coef=0.01, initial_lr=1e-2, norm_adv=False,
                              continuous_action=True)
    alg.grad_hook = pdist.FlatGradAllReduce(alg.model, average=True)
    b = _ppo_batch()
    sl = slice(rank * 8, rank * 8 + 8)
    for _ in range(3):
        alg.learn(b['obs'][sl], b['act'][sl], b['val'][sl], b['ret'][sl], b['logp'][sl], b['adv'][sl])
    pdist.barrier()
    q.put((rank, {k: v.numpy().copy() for k, v in alg.model.state_dict().items()}))


@pytest.mark.skipif(torch.cuda.is_available(), reason='PPO moves its model to the GPU when one is present (ppo.py:75-76)')
def test_ppo_data_parallel_mean_gradient(tmp_path):
    import parl_amd as parl
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_ppo_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sd0, sd1 = res[0][1], res[1][1]
    for k in sd0:
        assert np.array_equal(sd0[k], sd1[k]), k
    torch.manual_seed(200)
    m = _ppo_model()
    alg = parl.algorithms.PPO(m, clip_param=0.2, entropy_coef=0.01, initial_lr=1e-2, norm_adv=False,
                              continuous_action=True)
    b = _ppo_batch()
    for _ in range(3):  # equal shards: the mean over the union = the mean of the per-rank means
        alg.learn(b['obs'], b['act'], b['val'], b['ret'], b['logp'], b['adv'])
    for k, v in alg.model.state_dict().items():
        np.testing.assert_allclose(sd0[k], v.numpy(), rtol=2e-5, atol=2e-6)


def test_missing_rank_makes_init_raise_within_the_timeout():
    """parl_amd.dist.init(timeout_s): a peer that never shows up is an exception naming rank / world / address after
    the timeout, not torch's 10-30 minute default (the reference has no collectives to hang in)"""
    import subprocess
    import time
    code = ('import sys; sys.path.insert(0, %r)\n'
            'from parl_amd import dist as pdist\n'
            'try:\n'
            '    pdist.init(backend="gloo", timeout_s=3)\n'
            'except RuntimeError as e:\n'
            '    print("RAISED", e); sys.exit(7)\n') % ROOT
    env = dict(os.environ, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    t0 = time.time()
    p = subprocess.run([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=120, text=True)
    assert p.returncode == 7 and 'rendezvous failed on rank 0 of 2' in p.stdout, p.stdout[-2000:]
    assert time.time() - t0 < 60


def test_captures_are_thread_local_once_a_process_group_exists():
    """parl_amd.dist.graph_capture_kwargs(): every hipGraph capture of the package (rollout segments, the A2C rollout,
    all forms of the update graph) takes it.  Without a group: torch's default; with one: capture_error_mode
    'thread_local' — ProcessGroupNCCL's watchdog thread polls its collectives with hipEventQuery and, in the default
    mode, aborts the process if it does so while a capture is open."""
    import subprocess
    code = ('import sys; sys.path.insert(0, %r)\n'
            'from parl_amd import dist as pdist\n'
            'assert pdist.graph_capture_kwargs() == {}, pdist.graph_capture_kwargs()\n'
            'pdist.init(backend="gloo", force=True, timeout_s=30)\n'
            'assert pdist.graph_capture_kwargs() == {"capture_error_mode": "thread_local"}\n'
            'print("OK")\n') % ROOT
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    p = subprocess.run([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=120, text=True)
    assert p.returncode == 0 and 'OK' in p.stdout, p.stdout[-2000:]
    src = open(os.path.join(ROOT, 'parl_amd', 'rollout.py')).read() + \
        open(os.path.join(ROOT, 'parl_amd', 'algorithms', 'impala', 'graphed.py')).read()
    import re
    caps = re.findall(r'torch\.cuda\.graph\(([^\n]*)', src)
    # (every capture call passes the kwargs on: directly, as `tl`, or through a `kw` that was updated with them)
    assert len(caps) >= 6 and all(('graph_capture_kwargs' in c) or ('**tl' in c) or ('**kw' in c) or ('thread_local' in c)
                                  for c in caps), caps
    assert 'kw.update(pdist.graph_capture_kwargs())' in src


def _cartpole_dp_worker(rank, world, port, q, updates):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from parl_amd import dist as pdist
    import test_cartpole_a2c as cp
    pdist.init(backend='gloo', timeout_s=120)
    torch.set_num_threads(1)
    torch.manual_seed(100 + rank)           # different initial weights per rank: the broadcast must fix it
    n_act = cp.CONFIG['actor_num'] // world   # this rank's share of the 4 actors, with the seeds the single-process run gives them
    cfg = dict(cp.CONFIG, actor_num=n_act, actor_seed0=1 + rank * n_act)
    learner = cp.Learner(cfg, cp._oracle_calc_gae)
    model = learner.agent.alg.model
    pdist.broadcast_model(model)
    learner.agent.alg.grad_hook = pdist.FlatGradAllReduce(model)   # A2C's losses are sums: SUM of the ranks' gradients
    sched = learner.agent.lr_scheduler
    step1 = sched.step
    sched.step = lambda step_num=1: step1(step_num=step_num * world)   # the schedule counts the UNION batch's rows
    recent, curve = [], []
    for u in range(updates):
        losses = learner.step()
        assert np.isfinite(losses).all()
        recent = (recent + learner.episode_returns())[-20:]
        if (u + 1) % 20 == 0 and recent:
            curve.append((u + 1, float(np.mean(recent))))
    w = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    q.put((rank, float(np.mean(recent)), curve, float(w.double().sum()), float(w.abs().max())))


def test_two_rank_data_parallel_training_learns_cartpole():
    """Data-parallel semantics exercised as TRAINING, not only as gradient equality: configs[0]'s CartPole A2C
    (tests/test_cartpole_a2c.py: 4 remote actors x 4 envs, one A2C.learn per 320-row batch) split over two gloo
    ranks — 2 actors each, one SUM all-reduce of the flat gradient per update, clip on the reduced gradient, the
    lr schedule stepped by the union batch's rows.  Both ranks hold identical parameters after 140 updates and the
    episodes THEIR actors close pass a windowed mean return of 100 (measured: 161 / 172 at update 120; the
    single-process run reaches 150 after 113)."""
    world, updates = 2, 140
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cartpole_dp_worker, args=(r, world, port, q, updates)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.__stdout__.write('\nCartPole A2C, 2 DP ranks x 2 actors x 4 envs (update, mean return of the last 20 episodes): %s\n'
                         % [r[2] for r in res])
    assert res[0][3] == res[1][3] and res[0][4] == res[1][4], 'replicas diverged'
    assert all(max(v for _, v in r[2]) >= 100.0 for r in res), [r[2] for r in res]   # from ~20 of a random policy
