"""The example twins (examples/IMPALA, examples/A2C) run end to end on the device path, and the
on-device A2C rollout reproduces the reference's per-segment calc_gae semantics.  -m gpu."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(args, timeout=240):
    env = dict(os.environ, PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable] + args, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=timeout, text=True)
    assert p.returncode == 0, p.stdout[-3000:]
    return p.stdout


def test_a2c_example_runs():
    out = _run(['examples/A2C/train.py', '--max_sample_steps', '1600', '--env-num', '16', '--log-interval', '1'])
    assert "'sample_steps': " in out and "'total_loss': " in out


def test_impala_example_runs():
    """default mode: AsyncActorLearner with hipGraph updates of train_batch_size rows"""
    out = _run(['examples/IMPALA/train.py', '--minutes', '0.2', '--env-num', '16', '--train-batch-size', '400',
                '--log-interval', '3'])
    assert "'learn_steps': " in out and "'kl': " in out and "'learner_updates_per_s': " in out


def test_impala_example_runs_at_the_north_star_frame_size():
    """--env-dim 84: AtariModel84 (the 84x84 MFMA trunk kernels, heads by rocBLAS + the one-kernel V-trace loss)
    under the same pipeline; 3 minutes of it on 1024 envs: profiles/r03_impala_pong_84_3min.log"""
    out = _run(['examples/IMPALA/train.py', '--env-dim', '84', '--minutes', '0.2', '--env-num', '16',
                '--train-batch-size', '400', '--log-interval', '3'])
    assert "'learn_steps': " in out and "'kl': " in out and "'learner_updates_per_s': " in out


def test_impala_example_runs_with_the_reference_thread_structure():
    """--threads: class Learner (learn thread + queue + one sampling thread per @parl.remote_class Actor)"""
    out = _run(['examples/IMPALA/train.py', '--threads', '--minutes', '0.2', '--env-num', '16',
                '--train-batch-size', '400', '--log-interval', '3'])
    assert "'learn_steps': " in out and "'kl': " in out


def test_a2c_rollout_matches_per_segment_calc_gae(dev, oracle):
    """DeviceA2CRollout's one batched GAE launch == the reference's per-(env, segment) calc_gae with
    next_value = 0 after a terminal step (examples/A2C/actor.py:73-85), on a real Breakout rollout
    (life losses give mid-rollout dones)."""
    import parl_amd as parl
    from parl_amd.env import DeviceVectorEnv
    from parl_amd.models import AtariModel84
    from parl_amd.rollout import DeviceA2CRollout
    torch.manual_seed(0)
    E, T = 12, 40
    env = DeviceVectorEnv('BreakoutNoFrameskip-v4', E, dim=84, horizon=T, seed=4, device=dev)
    model = AtariModel84(env.act_dim).to(dev)
    ro = DeviceA2CRollout(env, T, gamma=0.99, lam=0.95, seed=2)
    for _ in range(3):  # a few rollouts so that some envs lose lives
        b = ro.collect(model)
    rew, val, done = ro.rewards.cpu().numpy(), ro.values.cpu().numpy(), ro.dones.cpu().numpy()
    with torch.no_grad():
        nv = model.value(env.current_obs()).cpu().numpy()
    adv = b['advantages'].reshape(T, E).cpu().numpy()
    tgt = b['target_values'].reshape(T, E).cpu().numpy()
    assert done.sum() > 0
    # float64 restatement of calc_gae per segment (rl_utils.py:34-51)
    for e in range(E):
        start = 0
        for t in range(T):
            if done[t, e] or t == T - 1:
                r, v = rew[start:t + 1, e].astype(np.float64), val[start:t + 1, e].astype(np.float64)
                nxt = 0.0 if done[t, e] else float(nv[e])
                td = r + 0.99 * np.append(v[1:], nxt) - v
                a = np.zeros_like(td)
                acc = 0.0
                for i in range(len(td) - 1, -1, -1):
                    acc = td[i] + 0.99 * 0.95 * acc
                    a[i] = acc
                np.testing.assert_allclose(adv[start:t + 1, e], a, rtol=1e-5, atol=1e-5)
                np.testing.assert_allclose(tgt[start:t + 1, e], a + v, rtol=1e-5, atol=1e-5)
                start = t + 1
    # the learner consumes it
    alg = parl.algorithms.A2C(model, vf_loss_coeff=0.5)
    total, pi, vf, ent = alg.learn(b['obs'], b['actions'], b['advantages'], b['target_values'], 1e-4, -0.01)
    assert np.isfinite(float(total))


def test_a2c_rollout_as_one_graph_equals_eager_launches(dev, monkeypatch):
    """DeviceA2CRollout replays a rollout as ONE hipGraph from its third call on (first eager, second captured):
    the same batches as eager launches, rollout after rollout, also after the weights changed in between (the
    model's operand-order weight buffers are refreshed inside the graph)"""
    import parl_amd as parl
    from parl_amd.env import DeviceVectorEnv
    from parl_amd.models import AtariModel84
    from parl_amd.rollout import DeviceA2CRollout
    E, T = 10, 8
    outs = []
    for graph in (True, False):
        monkeypatch.setenv('PARL_AMD_A2C_GRAPH', '1' if graph else '0')
        torch.manual_seed(4)
        env = DeviceVectorEnv('BreakoutNoFrameskip-v4', E, dim=84, horizon=T, seed=3, device=dev, max_episode_steps=700)
        model = AtariModel84(env.act_dim).to(dev)
        alg = parl.algorithms.A2C(model, vf_loss_coeff=0.5)
        ro = DeviceA2CRollout(env, T, gamma=0.99, lam=0.95, seed=6)
        assert ro._can_graph(model) == graph
        got = []
        for it in range(6):
            b = ro.collect(model)
            got.append({k: v.clone() for k, v in b.items()})
            if it in (1, 3):   # an update between rollouts: new weights under the same graph
                alg.learn(b['obs'], b['actions'], b['advantages'], b['target_values'], 3e-4, -0.01)
        got.append({'ep': ro.ep_stats.clone(), 'values': ro.values.clone()})
        assert (len(ro._graphs) == 1) == graph
        env.check_faults()
        outs.append(got)
    for i, (a, b) in enumerate(zip(*outs)):
        for k in a:
            assert torch.equal(a[k], b[k]), (i, k)


def test_a2c_rollout_rows_are_aligned(dev, monkeypatch):
    """Row (t, e) of the batch DeviceA2CRollout hands to A2C.learn carries the stacked observation the policy
    saw at step t for env e, the action drawn from THAT forward pass and the value it produced (the sums of
    a2c.py:67-79 do not care about the order of the rows, but they do care that a row's obs, action, advantage
    and target belong together).  The observations are recorded at the model call, the batch is compared
    with the recording after the rollout; several rollouts so that the ring has been rolled and the
    picture moves."""
    from parl_amd.env import DeviceVectorEnv
    from parl_amd.models import AtariModel84
    from parl_amd.rollout import DeviceA2CRollout
    monkeypatch.setenv('PARL_AMD_A2C_GRAPH', '0')   # the recording below hooks the model call: eager launches
    torch.manual_seed(1)
    E, T = 6, 12
    env = DeviceVectorEnv('PongNoFrameskip-v4', E, dim=84, horizon=T, seed=7, device=dev)
    model = AtariModel84(env.act_dim).to(dev)
    ro = DeviceA2CRollout(env, T, gamma=0.99, lam=1.0, seed=5)
    seen, outs = [], []
    inner = model.policy_and_value

    def recording(obs):
        logits, values = inner(obs)
        # (the model reads the frame ring in place: what it saw is the materialised stack of that reference)
        seen.append(obs.materialize().clone() if hasattr(obs, 'materialize') else obs.clone())
        outs.append((logits.clone(), values.clone()))
        return logits, values

    model.policy_and_value = recording
    moved = False
    for it in range(4):
        del seen[:], outs[:]
        b = ro.collect(model)
        torch.cuda.synchronize()
        assert len(seen) == T
        obs = b['obs'].reshape(T, E, 4, 84, 84)
        act = b['actions'].reshape(T, E)
        val = (b['target_values'] - b['advantages']).reshape(T, E)
        for t in range(T):
            assert torch.equal(obs[t], seen[t]), (it, t)
            torch.testing.assert_close(val[t], outs[t][1], rtol=0, atol=1e-5)
            assert torch.equal(act[t], ro.actions[t])
            assert int(act[t].min()) >= 0 and int(act[t].max()) < env.act_dim
        # the newest frame of step t is the second newest of step t + 1 (FrameStack), inside one episode
        for t in range(T - 1):
            keep = ro.dones[t] == 0
            assert torch.equal(obs[t + 1][keep][:, 2], obs[t][keep][:, 3])
        moved = moved or not torch.equal(obs[0], obs[T - 1])
    assert moved, 'the picture never changed: the comparison above proved nothing'


def test_async_actor_learner_pipeline(dev):
    """AsyncActorLearner (learner update on batch i-1 overlapped with the collection of batch i on
    a second stream): the first batch is bit-identical to a plain DeviceRollout's, the behaviour
    policy lags the learner by exactly one update, the two trajectory buffers alternate."""
    import copy
    import parl_amd as parl
    from parl_amd.env import DeviceVectorEnv
    from parl_amd.models import AtariModel42
    from parl_amd.rollout import AsyncActorLearner, DeviceRollout
    torch.manual_seed(0)
    E, T = 16, 8
    model = AtariModel42(6).to(dev)
    ref_model = copy.deepcopy(model)
    mk = lambda: DeviceVectorEnv('PongNoFrameskip-v4', E, dim=42, horizon=T, seed=11, device=dev)
    alg = parl.algorithms.IMPALA(model, sample_batch_steps=T, gamma=0.99, vf_loss_coeff=0.5,
                                 clip_rho_threshold=1.0, clip_pg_rho_threshold=1.0)
    pipe = AsyncActorLearner(alg, mk(), T, seed=5)
    pipe.prime()
    pipe.synchronize()
    b0 = {k: v.clone() for k, v in pipe.pending[0][0].items()}
    ref = DeviceRollout(mk(), T, seed=5).collect(ref_model)
    for k in ref:
        assert torch.equal(b0[k], ref[k]), k
    used = [pipe.pending[1]]
    for i in range(3):
        before = [p.detach().clone() for p in model.parameters()]
        loss, kl = pipe.step(1e-3, -0.01)
        used.append(pipe.pending[1])
        pipe.synchronize()
        assert np.isfinite(float(loss.total_loss.item()))
        # the rollout enqueued by this step() acted with the parameters from BEFORE this update
        for a, b in zip(pipe.actor_model.parameters(), before):
            assert torch.equal(a, b)
        assert any(not torch.equal(p, b) for p, b in zip(model.parameters(), before))
    assert used == [0, 1, 0, 1]
    pipe.env.check_faults()


def test_async_actor_learner_two_groups_equals_one_vector(dev):
    """Two env groups on two actor streams produce exactly the trajectories of ONE vector env of
    all envs (same env ids => same Philox streams), and learn_batches() on the two group batches
    gives the same update as learn() on the union batch."""
    import copy
    import parl_amd as parl
    from parl_amd.env import DeviceVectorEnv
    from parl_amd.models import AtariModel42
    from parl_amd.rollout import AsyncActorLearner, DeviceRollout
    torch.manual_seed(0)
    E, T = 16, 8
    model = AtariModel42(6).to(dev)
    model_b = copy.deepcopy(model)
    mk = lambda n, id0: DeviceVectorEnv('PongNoFrameskip-v4', n, dim=42, horizon=T, seed=11, env_id0=id0, device=dev)
    kw = dict(sample_batch_steps=T, gamma=0.99, vf_loss_coeff=0.5, clip_rho_threshold=1.0, clip_pg_rho_threshold=1.0)
    alg = parl.algorithms.IMPALA(model, **kw)
    pipe = AsyncActorLearner(alg, [mk(E // 2, 0), mk(E // 2, E // 2)], T, seed=5)
    pipe.prime()
    pipe.synchronize()
    ref = DeviceRollout(mk(E, 0), T, seed=5).collect(model_b)
    ga, gb = pipe.pending[0]
    for k in ref:
        full = ref[k].reshape((T, E) + tuple(ref[k].shape[1:]))
        assert torch.equal(ga[k].reshape((T, E // 2) + tuple(ga[k].shape[1:])), full[:, :E // 2]), k
        assert torch.equal(gb[k].reshape((T, E // 2) + tuple(gb[k].shape[1:])), full[:, E // 2:]), k
    # one update on the union == accumulated update on the two halves
    alg_b = parl.algorithms.IMPALA(model_b, **kw)
    la, _ = alg.learn_batches([{k: v.clone() for k, v in ga.items()}, {k: v.clone() for k, v in gb.items()}],
                              1e-3, -0.01, time_major=True)
    lb, _ = alg_b.learn(ref['obs'], ref['actions'], ref['behaviour_logits'], ref['rewards'], ref['dones'], 1e-3,
                        -0.01, time_major=True)
    np.testing.assert_allclose(float(la.total_loss), float(lb.total_loss), rtol=1e-5)
    for pa, pb in zip(model.parameters(), model_b.parameters()):
        np.testing.assert_allclose(pa.detach().cpu().numpy(), pb.detach().cpu().numpy(), rtol=2e-4, atol=2e-6)


def test_bench_two_ranks_share_one_gpu(tmp_path):
    """bench.py's N>1 path (env sharding by rank, weight broadcast, flat-gradient all-reduce inside
    the async actor/learner pipeline, trajectory all-gather, max-over-ranks timing) with two ranks
    on the one GPU of the test box.  RCCL refuses two ranks per device, so the collectives go over
    gloo here; the code path above the backend is the one the 8-GPU launch runs."""
    import json
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, PYTHONPATH=ROOT, PARL_AMD_SHARE_GPU='1', PARL_AMD_DIST_BACKEND='gloo',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', str(port), 'bench.py', '--gpus', '2', '--steps', '2', '--warmup', '1',
           '--envs', '64', '--sample-batch-steps', '10', '--no-cpu-baseline']
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=400, text=True)
    assert p.returncode == 0, p.stdout[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(line) == 1, p.stdout[-2000:]
    out = json.loads(line[0])
    assert out['n_gpus'] == 2 and out['scaling'] == 'weak' and out['value'] > 0
    assert out['config']['train_batch'] == 2 * 64 * 10


def test_bench_eight_ranks_dry_run_on_one_gpu():
    """The real rank count of BASELINE configs[3] (8 ranks), self-launched by `python bench.py --gpus 8` exactly as
    the driver launches it (torch.distributed.run, one process per rank), on the ONE GPU of the test box over
    gloo: env-id ranges per rank, rendezvous / port handling, weight broadcast, the order of collectives (flat-
    gradient all-reduce, small-tensor all-gather, max-over-ranks timing, barriers) with 8 participants.  Not a
    scaling number (the JSON line says so)."""
    import json
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('WORLD_SIZE', None)
    cmd = [sys.executable, 'bench.py', '--gpus', '8', '--envs', '32', '--quick', '--steps', '2', '--warmup', '1',
           '--sample-batch-steps', '10', '--no-cpu-baseline']
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, text=True)
    assert p.returncode == 0, '\n'.join(ln for ln in p.stdout.splitlines() if 'amdgpu.ids' not in ln and 'hostname' not in ln)[-12000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(line) == 1, p.stdout[-2000:]
    out = json.loads(line[0])
    assert out['n_gpus'] == 8 and out['value'] > 0 and out['config']['train_batch'] == 8 * 32 * 10
    assert 'SHARE' in out['config']['collectives']
    assert out['config']['env_ids_per_rank'] == [[r * 32, r * 32 + 31] for r in range(8)]


def test_bench_two_ranks_reference_train_batch(tmp_path):
    """the data-parallel form of the hipGraph updates (train_batch_size mode): forward + backward graph | gradient
    all-reduce | clip + Adam graph, two ranks sharing the GPU over gloo"""
    import json
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('WORLD_SIZE', None)
    cmd = [sys.executable, 'bench.py', '--gpus', '2', '--envs', '16', '--quick', '--steps', '2', '--warmup', '1',
           '--sample-batch-steps', '10', '--train-batch', '40', '--no-cpu-baseline']
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, text=True)
    assert p.returncode == 0, p.stdout[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    out = json.loads(line[0])
    assert out['n_gpus'] == 2 and out['config']['learner_updates_per_step'] == 4  # 16 sequences / (40 // 10)
    assert out['learner_updates_per_sec'] > 0


def test_ppo_example_atari_runs():
    out = _run(['examples/PPO/train.py', '--env_num', '8', '--step_nums', '32', '--train_total_steps', '512'])
    assert "'value_loss': " in out and "'update': 2" in out


def test_ppo_example_continuous_runs():
    out = _run(['examples/PPO/train.py', '--continuous_action', '--env_num', '64', '--step_nums', '64',
                '--train_total_steps', '8192'])
    assert "'action_loss': " in out and "'update': 2" in out


def test_env_and_sampler_checkpoint_resumes_bit_identically(dev, tmp_path):
    """Agent.save with the env and the sampler attached, 20 more steps, Agent.restore,
    the same 20 steps again: observations, actions, rewards, dones identical — env state blobs,
    frame ring, `since`, reset counters and the Philox offsets all come back.  Breakout, so that
    life losses / FIRE resets / noop draws happen inside the window."""
    import parl_amd as parl
    from parl_amd.env import DeviceVectorEnv
    from parl_amd.models import AtariModel42
    from parl_amd.rollout import DeviceRollout
    torch.manual_seed(0)
    E, T = 24, 20
    env = DeviceVectorEnv('BreakoutNoFrameskip-v4', E, dim=42, horizon=T, seed=11, device=dev)
    model = AtariModel42(env.act_dim).to(dev)
    alg = parl.algorithms.IMPALA(model, sample_batch_steps=T, gamma=0.99, vf_loss_coeff=0.5,
                                 clip_rho_threshold=1.0, clip_pg_rho_threshold=1.0)

    class A(parl.Agent):
        pass

    agent = A(alg)
    ro = DeviceRollout(env, T, seed=5)
    agent.attach(env=env, rollout=ro)
    for _ in range(4):  # get into the game
        ro.collect(model)
    path = str(tmp_path / 'ckpt' / 'model.ckpt')
    agent.save(path)
    assert os.path.exists(path) and os.path.exists(path + '.env')

    def run():
        b = ro.collect(model)
        return [b[k].clone() for k in ('obs', 'actions', 'rewards', 'dones', 'behaviour_logits')]

    first = run()
    second_without_restore = run()
    assert not torch.equal(first[0], second_without_restore[0])
    agent.restore(path)
    again = run()
    for a, b in zip(first, again):
        assert torch.equal(a, b)
    assert int(first[3].sum()) > 0  # dones (life losses) inside the window
    # a different env cannot swallow the checkpoint
    other = DeviceVectorEnv('BreakoutNoFrameskip-v4', E, dim=42, horizon=T, seed=12, device=dev)
    with pytest.raises(ValueError):
        other.load_state_dict(env.state_dict())


def test_async_pipeline_checkpoint_resumes_bit_identically(dev, tmp_path):
    """AsyncActorLearner.state_dict / load_state_dict (envs and rings written on the ACTOR stream, launches ahead
    of the host; the collected-but-unlearned `pending` batch; the actors' weight snapshot): save mid-run, three
    more steps, restore, the same three steps again — every parameter and the pending batch bit-identical."""
    import parl_amd as parl
    from parl_amd.env import DeviceVectorEnv
    from parl_amd.models import AtariModel42
    from parl_amd.rollout import AsyncActorLearner
    torch.manual_seed(0)
    E, T = 16, 8
    env = DeviceVectorEnv('BreakoutNoFrameskip-v4', E, dim=42, horizon=T, seed=21, device=dev)
    model = AtariModel42(env.act_dim).to(dev)
    with torch.no_grad():
        model.policy_fc.weight.mul_(0.05)
        model.value_fc.weight.mul_(0.05)
    alg = parl.algorithms.IMPALA(model, sample_batch_steps=T, gamma=0.99, vf_loss_coeff=0.5,
                                 clip_rho_threshold=1.0, clip_pg_rho_threshold=1.0)

    class A(parl.Agent):
        pass

    agent = A(alg)
    pipe = AsyncActorLearner(alg, [env], T, seed=7)
    agent.attach(pipeline=pipe)
    for _ in range(3):
        pipe.step(1e-3, -0.01)
    path = str(tmp_path / 'async.ckpt')
    agent.save(path)  # no synchronize() by the caller: state_dict drains the streams itself
    opt_state = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in
                 {'state': [{n: t.clone() for n, t in st.items()} for st in alg.optimizer.state_dict()['state'].values()]}.items()}
    full_opt = alg.optimizer.state_dict()
    full_opt = {'state': {k: {n: (t.clone() if isinstance(t, torch.Tensor) else t) for n, t in st.items()}
                          for k, st in full_opt['state'].items()}, 'param_groups': full_opt['param_groups']}
    del opt_state

    def run():
        for _ in range(3):
            pipe.step(1e-3, -0.01)
        pipe.synchronize()
        return ([p.detach().clone() for p in model.parameters()],
                {k: v.clone() for k, v in pipe.pending[0][0].items()})

    p1, b1 = run()
    p_drift, _ = run()
    assert any(not torch.equal(a, b) for a, b in zip(p1, p_drift))
    agent.restore(path)
    alg.optimizer.load_state_dict(full_opt)
    p2, b2 = run()
    for a, b in zip(p1, p2):
        assert torch.equal(a, b)
    for k in b1:
        assert torch.equal(b1[k], b2[k]), k
    env.check_faults()
    # a model-only checkpoint cannot silently restart attached envs
    os.remove(path + '.env')
    with pytest.raises(FileNotFoundError):
        agent.restore(path)


def test_rollout_checkpoint_names_its_observation_layout(dev):
    """a checkpoint written with the observations in the env's frame rings (lazy_obs) cannot be resumed into a
    rollout that materialises them per buffer (or the other way round) — the mismatch is a descriptive error, not a
    bare KeyError / a pending batch that silently reads whichever ring is current"""
    from parl_amd.env import DeviceVectorEnv
    from parl_amd.models import AtariModel42
    from parl_amd.rollout import DeviceRollout
    E, T = 8, 6
    mk = lambda lazy: DeviceRollout(DeviceVectorEnv('PongNoFrameskip-v4', E, dim=42, horizon=T, seed=3, device=dev), T,
                                    seed=1, n_buffers=2, lazy_obs=lazy)
    model = AtariModel42(6).to(dev)
    lazy, plain = mk(True), mk(False)
    lazy.collect(model)
    plain.collect(model)
    d_lazy, d_plain = lazy.state_dict(), plain.state_dict()
    assert d_lazy['lazy_obs'] is True and d_plain['lazy_obs'] is False
    with pytest.raises(ValueError, match='lazy_obs'):
        plain.load_state_dict(d_lazy)
    with pytest.raises(ValueError, match='lazy_obs'):
        lazy.load_state_dict(d_plain)
    with pytest.raises(ValueError, match="holds no 'obs'"):
        plain.load_buffer_state(0, lazy.buffer_state(0))
    bad = dict(d_lazy, ring_of_buf=[0])
    with pytest.raises(ValueError, match='frame ring'):
        lazy.load_state_dict(bad)
    lazy.load_state_dict(d_lazy)   # the matching layout loads
