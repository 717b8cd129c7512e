"""Elastic launches (parlhip_atari_vec_step_elastic, ElasticDeviceRollout): an env inside a life-loss
reset drops out of the next launches instead of holding the whole vector up.  What must NOT change is
every env's own trajectory: the rows of an elastic batch are replayed action by action through the
CPU oracle's synchronous VectorEnv (and, at a larger size, through the device's synchronous path) and
must be bit-identical — observations, rewards, dones, episode records.  Needs a real MI355X: -m gpu."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BREAKOUT = 'BreakoutNoFrameskip-v4'


def _rom(game):
    from parl_amd.env import find_rom
    try:
        return find_rom(game)
    except FileNotFoundError:
        pytest.skip('cartridge %s.bin not present' % game)


def _model(dev, act_dim, seed=0):
    from parl_amd.models.atari_model import AtariModel42
    torch.manual_seed(seed)
    m = AtariModel42(act_dim).to(dev)
    with torch.no_grad():  # near-uniform behaviour policy: FIRE often enough to lose lives quickly
        m.policy_fc.weight.mul_(0.05)
        m.policy_fc.bias.zero_()
    return m


def _elastic(dev, game, E, T, dim, seed, rom, env_id0=0, fused_obs=None):
    from parl_amd.env import DeviceVectorEnv
    from parl_amd.rollout import ElasticDeviceRollout
    env = DeviceVectorEnv(game, E, dim=dim, horizon=4 * T + 32, seed=seed, env_id0=env_id0, device=dev, rom_bytes=rom)
    if fused_obs is not None:   # the observation at the tail of the env launch (default) or as a frame_post launch behind it
        assert env.fused_obs or not fused_obs
        env.fused_obs = bool(fused_obs)
    return env, ElasticDeviceRollout(env, T, seed=seed + 1)


@pytest.mark.parametrize('fused_obs', [True, False])
def test_elastic_rows_replay_through_the_oracle(dev, oracle, fused_obs):
    E, T, dim, seed, batches = 16, 10, 42, 3, 14
    rom = _rom('breakout')
    env, ro = _elastic(dev, BREAKOUT, E, T, dim, seed, rom, fused_obs=fused_obs)
    model = _model(dev, env.act_dim)
    orc = oracle.VecEnv(rom, 'breakout', E, dim, seed=seed)
    o_prev = orc.reset()
    # MonitorEnv records of every launch, per env in launch order
    dev_eps = [[] for _ in range(E)]
    acc = env.accumulate_episode_stats

    def recording(acc3):
        ln = env.ep_lengths.cpu().numpy()
        rt = env.ep_returns.cpu().numpy()
        for e in np.nonzero(ln)[0]:
            dev_eps[e].append((float(rt[e]), int(ln[e])))
        return acc(acc3)

    env.accumulate_episode_stats = recording
    total = ndone = 0
    orc_eps = [[] for _ in range(E)]
    for b in range(batches):
        batch = ro.collect(model)
        torch.cuda.synchronize()
        total += ro.launches
        obs = batch['obs'].view(T, E, 4, dim, dim).cpu().numpy()
        act = batch['actions'].view(T, E).cpu().numpy()
        rew = batch['rewards'].view(T, E).cpu().numpy()
        don = batch['dones'].view(T, E).cpu().numpy()
        h = (b & 1) * T
        rl = ro.row_launch[h:h + T].cpu().numpy()  # launches since the reset, per (row, env)
        assert (np.diff(rl, axis=0) >= 1).all() and rl.max() < ro.launch and rl.min() >= 0
        # the policy output of the row's launch travels with the row
        lm = ro.logits_lm.cpu().numpy()
        bl = batch['behaviour_logits'].view(T, E, -1).cpu().numpy()
        for e in range(E):
            assert np.array_equal(bl[:, e], lm[rl[:, e] % ro.S, e])
        for r in range(T):
            assert np.array_equal(obs[r], o_prev), 'obs, batch %d row %d' % (b, r)
            o_prev, orr, od = orc.step(act[r])
            assert np.array_equal(rew[r], orr), 'reward, batch %d row %d' % (b, r)
            assert np.array_equal(don[r].astype(np.uint8), od), 'done, batch %d row %d' % (b, r)
            ndone += int(od.sum())
        for e in range(E):
            orc_eps[e] += orc.pop_episodes(e)
    env.check_faults()
    # fast envs are up to one batch ahead of the rows replayed: the oracle's records are a prefix of the device's
    n_orc = 0
    for e in range(E):
        assert dev_eps[e][:len(orc_eps[e])] == orc_eps[e], e
        n_orc += len(orc_eps[e])
    n, r, l = ro.pop_episode_stats()
    assert n == sum(len(x) for x in dev_eps) >= n_orc
    assert ndone >= E, 'test too short: %d life losses' % ndone
    assert total > T * batches, 'no launch was ever elastic'
    # envs ran ahead: when the last batch closed some had already started rows of the next one
    assert int(ro.rows_done.max()) > T * batches


def test_elastic_equals_synchronous_device_rollout_at_size(dev):
    """256 envs, 12 batches: every row of every env == the synchronous device path fed the same actions"""
    from parl_amd.env import DeviceVectorEnv
    E, T, dim, seed, batches = 256, 16, 42, 11, 12
    rom = _rom('breakout')
    env, ro = _elastic(dev, BREAKOUT, E, T, dim, seed, rom, env_id0=512)
    model = _model(dev, env.act_dim, seed=1)
    ref = DeviceVectorEnv(BREAKOUT, E, dim=dim, horizon=8, seed=seed, env_id0=512, device=dev, rom_bytes=rom)
    o_prev = ref.reset().clone()
    launches = 0
    for b in range(batches):
        batch = ro.collect(model)
        launches += ro.launches
        obs = batch['obs'].view(T, E, 4, dim, dim)
        act = batch['actions'].view(T, E)
        for r in range(T):
            assert torch.equal(obs[r], o_prev), 'obs, batch %d row %d' % (b, r)
            o, rr, dd, _ = ref.step(act[r].contiguous())
            o_prev = o.clone()
            assert torch.equal(batch['rewards'].view(T, E)[r], rr)
            assert torch.equal(batch['dones'].view(T, E)[r], dd)
    env.check_faults()
    ref.check_faults()
    # the point of it: far fewer than the 4 launch-times per row a synchronous vector pays when one env resets
    assert T * batches < launches < 1.35 * T * batches


def test_elastic_pong_is_the_synchronous_rollout(dev):
    """a game without lives never suspends: exactly T launches, same batch as DeviceRollout bit for bit"""
    from parl_amd.env import DeviceVectorEnv
    from parl_amd.rollout import DeviceRollout, ElasticDeviceRollout
    E, T, dim = 64, 8, 42
    rom = _rom('pong')
    model = _model(dev, 6)
    e1 = DeviceVectorEnv('PongNoFrameskip-v4', E, dim=dim, horizon=4 * T + 32, seed=5, device=dev, rom_bytes=rom)
    e2 = DeviceVectorEnv('PongNoFrameskip-v4', E, dim=dim, horizon=T, seed=5, device=dev, rom_bytes=rom)
    # poll_lag 0: the host waits for the completion counter after every launch from T-1 on, so no idle
    # launch is enqueued past the end and the Philox offsets (one per launch) stay those of DeviceRollout
    r1, r2 = ElasticDeviceRollout(e1, T, seed=9, poll_lag=0), DeviceRollout(e2, T, seed=9)
    for _ in range(3):
        b1, b2 = r1.collect(model), r2.collect(model)
        assert r1.launches == T
        for k in b2:
            assert torch.equal(b1[k], b2[k]), k
    # the default lagged poll: the poll_lag launches enqueued past the closing one already work on the next batch
    e3 = DeviceVectorEnv('PongNoFrameskip-v4', E, dim=dim, horizon=4 * T + 32, seed=5, device=dev, rom_bytes=rom)
    r3 = ElasticDeviceRollout(e3, T, seed=9)
    r3.collect(model)
    assert r3.launches == T + r3.poll_lag
    r3.collect(model)
    assert r3.launches == T


def test_async_actor_learner_elastic(dev):
    from parl_amd.algorithms import IMPALA
    from parl_amd.env import DeviceVectorEnv
    from parl_amd.models.atari_model import AtariModel42
    from parl_amd.rollout import AsyncActorLearner
    E, T = 64, 10
    rom = _rom('breakout')
    env = DeviceVectorEnv(BREAKOUT, E, dim=42, horizon=4 * T + 32, seed=0, device=dev, rom_bytes=rom)
    torch.manual_seed(0)
    alg = IMPALA(AtariModel42(env.act_dim).to(dev), sample_batch_steps=T, gamma=0.99, vf_loss_coeff=0.5,
                 clip_rho_threshold=1.0, clip_pg_rho_threshold=1.0)
    aal = AsyncActorLearner(alg, env, T, seed=0, elastic=True)
    for _ in range(12):
        loss, kl = aal.step(1e-4, -0.01)
    aal.synchronize()
    assert np.isfinite(float(loss.total_loss)) and np.isfinite(float(kl))
    assert aal.rollout.launch > 12 * T
    env.check_faults()


def test_elastic_checkpoint_resumes_bit_identically(dev):
    """env blobs (incl. parked wrapper state machines), the linked frame ring and the rollout's row tables
    survive state_dict / load_state_dict in fresh objects: the next batches are bit-identical"""
    E, T, dim, seed = 48, 12, 42, 21
    rom = _rom('breakout')
    model = _model(dev, 4)
    env, ro = _elastic(dev, BREAKOUT, E, T, dim, seed, rom)
    for _ in range(6):
        ro.collect(model)
    torch.cuda.synchronize()
    snap_env, snap_ro = env.state_dict(), ro.state_dict()
    want = [{k: v.clone() for k, v in ro.collect(model).items()} for _ in range(4)]
    env2, ro2 = _elastic(dev, BREAKOUT, E, T, dim, seed, rom)
    env2.load_state_dict(snap_env)
    ro2.load_state_dict(snap_ro)
    for w in want:
        got = ro2.collect(model)
        for k in w:
            assert torch.equal(got[k], w[k]), k
    env2.check_faults()
