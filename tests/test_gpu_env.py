"""Device Atari env (HIP emulator + wrapper state machine + frame_post + frame-stack ring, through
the C ABI) against the CPU oracle: bit-exact RAM, raw frames, observations, rewards, dones and
MonitorEnv episode records.  Needs a real MI355X: -m gpu."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu

GAMES = {'pong': 'PongNoFrameskip-v4', 'breakout': 'BreakoutNoFrameskip-v4'}


def _rom(game):
    from parl_amd.env import find_rom
    try:
        return find_rom(game)
    except FileNotFoundError:
        pytest.skip('cartridge %s.bin not present' % game)


def _run_parity(dev, oracle, game, E, dim, steps, seed, max_episode_steps=400000, cache=True, check_ram=True,
                fused=None):
    from parl_amd.env import DeviceVectorEnv
    rom = _rom(game)
    env = DeviceVectorEnv(GAMES[game], E, dim=dim, horizon=8, seed=seed, device=dev, rom_bytes=rom,
                          max_episode_steps=max_episode_steps, use_reset_cache=cache)
    if fused is not None:  # the one-launch step (parlhip_atari_vec_step_obs) or the two-launch form, explicitly
        assert env.fused_obs or not fused
        env.fused_obs = bool(fused)
    orc = oracle.VecEnv(rom, game, E, dim, seed=seed, max_episode_steps=max_episode_steps)
    assert np.array_equal(env.reset().cpu().numpy(), orc.reset())
    sb = env.states.numel() // E
    rng = np.random.default_rng(seed)
    ndone = 0
    for i in range(steps):
        a = rng.integers(0, env.act_dim, E)
        o, r, d, info = env.step(torch.from_numpy(a).to(dev))
        oo, orr, od = orc.step(a)
        assert np.array_equal(r.cpu().numpy(), orr), 'reward, step %d' % i
        assert np.array_equal(d.cpu().numpy().astype(np.uint8), od), 'done, step %d' % i
        assert np.array_equal(o.cpu().numpy(), oo), 'obs, step %d' % i
        if check_ram:
            ram = env.states.view(E, sb)[:, :128].cpu().numpy()
            for e in range(E):
                assert np.array_equal(ram[e], orc.ram(e)), 'ram env %d step %d' % (e, i)
        ln = info['episode_lengths'].cpu().numpy()
        rt = info['episode_returns'].cpu().numpy()
        for e in range(E):
            eps = orc.pop_episodes(e)
            if eps:
                assert (rt[e], ln[e]) == (eps[-1][0], eps[-1][1])
            else:
                assert ln[e] == 0
        ndone += int(od.sum())
    env.check_faults()
    return ndone


@pytest.mark.parametrize('game,dim', [('pong', 84), ('pong', 42), ('breakout', 84), ('breakout', 42)])
def test_env_matches_oracle(dev, oracle, game, dim):
    nd = _run_parity(dev, oracle, game, E=6, dim=dim, steps=160, seed=3)
    if game == 'breakout':
        assert nd > 0  # life losses / FIRE resets were exercised


@pytest.mark.parametrize('game,dim', [('pong', 84), ('breakout', 42)])
def test_env_matches_oracle_two_launch_form(dev, oracle, game, dim):
    """the default step is ONE launch (the observation at the tail of the env kernel); the two-launch form
    (parlhip_atari_vec_step, then parlhip_frame_post_step_u8) stays the path of other frame sizes / 4K cartridges"""
    _run_parity(dev, oracle, game, E=6, dim=dim, steps=60, seed=4, fused=False)


@pytest.mark.parametrize('game,dim,E', [('pong', 42, 9), ('pong', 84, 6), ('breakout', 42, 6), ('breakout', 84, 5)])
def test_step_obs_one_launch_equals_two_launches(dev, game, dim, E):
    """parlhip_atari_vec_step_obs against parlhip_atari_vec_step + parlhip_frame_post_step_u8 on twin envs: ring
    slot, FrameStack counters, rewards, dones, raw frames, state blobs and the MonitorEnv sums, every step"""
    from parl_amd.env import DeviceVectorEnv
    rom = _rom(game)
    mk = lambda: DeviceVectorEnv(GAMES[game], E, dim=dim, horizon=16, seed=11, device=dev, rom_bytes=rom,
                                 max_episode_steps=700)
    one, two = mk(), mk()
    assert one.fused_obs
    two.fused_obs = False
    acc1 = torch.zeros(3, dtype=torch.float64, device=dev)
    acc2 = torch.zeros(3, dtype=torch.float64, device=dev)
    assert torch.equal(one.reset(), two.reset())
    rng = np.random.default_rng(5)
    resets = 0
    for i in range(400):
        if one.t >= one.horizon:
            one.roll()
            two.roll()
        a = torch.from_numpy(rng.integers(0, one.act_dim, E)).to(dev)
        one.step_async(a, ep_acc=acc1)
        two.step_async(a, ep_acc=acc2)
        slot = one.t + 3
        assert torch.equal(one.ring[slot], two.ring[slot]), 'observation, step %d' % i
        assert torch.equal(one.since[slot], two.since[slot]), 'since, step %d' % i
        assert torch.equal(one.rewards, two.rewards) and torch.equal(one.dones, two.dones)
        assert torch.equal(one.obs_flags, two.obs_flags)
        assert torch.equal(one.raw_frames, two.raw_frames), 'raw frames, step %d' % i
        assert torch.equal(one.states, two.states), 'state blobs, step %d' % i
        resets += int((one.obs_flags & 2).ne(0).sum())
    assert torch.equal(acc1, acc2) and float(acc1[0]) > 0
    assert resets > 0
    one.check_faults()


def test_env_matches_oracle_timelimit_no_cache(dev, oracle):
    """TimeLimit + the never-reset CompatWrapper counter + the general (uncached) reset path"""
    assert _run_parity(dev, oracle, 'breakout', E=4, dim=42, steps=260, seed=8, max_episode_steps=600, cache=False) > 0
    assert _run_parity(dev, oracle, 'pong', E=4, dim=42, steps=200, seed=9, max_episode_steps=500, cache=True) > 0


def test_env_many_envs_ragged_block(dev, oracle):
    """E not a multiple of the 4 envs per workgroup; more envs than one wave per SIMD"""
    _run_parity(dev, oracle, 'pong', E=13, dim=42, steps=24, seed=21, check_ram=False)


def test_frame_post_rgb_and_colour_paths(dev, oracle):
    """frame_post on synthetic frames (SURVEY §8d 'frames' row): RGB input (the WarpFrame boundary)
    and TIA-colour input agree with the oracle bit-exactly, with and without the max."""
    import ctypes
    from parl_amd import _native as N
    rng = np.random.default_rng(0)
    E = 5
    L = N.lib()
    for dim in (84, 42):
        nb = L.parlhip_frame_post_tables_bytes(dim)
        blob = np.zeros(nb, np.uint8)
        N.check(L.parlhip_frame_post_tables_init(blob.ctypes.data, dim), 'tables')
        tab = torch.from_numpy(blob).to(dev)
        blocks = rng.integers(0, 128, (E, 2, 27, 20)).astype(np.uint8) * 2  # 8x8 blocks of palette colours
        col = np.repeat(np.repeat(blocks, 8, 2), 8, 3)[:, :, :210, :160].copy()
        col[:, :, 100:116, 40:44] = 0x0e  # a "paddle"
        pal = (ctypes.c_uint32 * 128)()
        oracle.lib().oracle_palette(pal)
        p = np.frombuffer(pal, np.uint32)
        rgbpal = np.stack([(p >> 16) & 255, (p >> 8) & 255, p & 255], -1).astype(np.uint8)
        rgb = rgbpal[col >> 1]  # [E,2,210,160,3]
        for two in (True, False):
            for fmt, src in ((1, col), (0, rgb)):
                f = torch.from_numpy(np.ascontiguousarray(src)).to(dev)
                per = f[0, 0].numel()
                out = torch.zeros((E, dim * dim), dtype=torch.uint8, device=dev)
                N.check(
                    L.parlhip_frame_post_u8(f.data_ptr(), (f.data_ptr() + per) if two else None, 2 * per, fmt, None,
                                            out.data_ptr(), dim * dim, E, dim, tab.data_ptr(), N.stream_ptr()),
                    'frame_post')
                ref = oracle.frame_post(src[:, 0], src[:, 1] if two else None, dim, fmt)
                assert np.array_equal(out.cpu().numpy().reshape(E, dim, dim), ref), (dim, two, fmt)


def test_stack_ring_matches_framestack_semantics(dev):
    """FrameStack.reset fills 4x the first frame (atari_wrappers.py:290-294); the ring reconstructs
    stacks by index from single frames."""
    from parl_amd.env import DeviceVectorEnv
    env = DeviceVectorEnv('PongNoFrameskip-v4', 3, dim=42, horizon=6, seed=1, device=dev, rom_bytes=_rom('pong'))
    o = env.reset()
    assert all(torch.equal(o[:, 0], o[:, j]) for j in range(1, 4))
    prev = o
    for i in range(5):
        o, r, d, _ = env.step(torch.zeros(3, dtype=torch.int64, device=dev))
        assert torch.equal(o[:, :3], prev[:, 1:])  # shift by one frame when no reset happened
        prev = o
    env.roll()
    assert torch.equal(env.current_obs(), prev)  # rolling the ring keeps the history


def test_conv12_reads_the_ring_like_the_gathered_stack(dev):
    """The actors' conv1 + conv2 on an ops.RingObservation (four single frames of the ring, FrameStack's repeat of
    the first frame after a reset through the `since` byte) against the same kernel on the materialised stack:
    bit-identical, over the steps after a reset (since = 0, 1, 2, 3), across episode ends and a ring roll; and the
    rollout's step with a ring-reading model draws the actions of the stack-reading one."""
    from parl_amd import ops
    from parl_amd.env import DeviceVectorEnv
    from parl_amd.models import AtariModel42
    from parl_amd.rollout import DeviceRollout
    E = 37
    env = DeviceVectorEnv('BreakoutNoFrameskip-v4', E, dim=42, horizon=12, seed=5, device=dev, rom_bytes=_rom('breakout'),
                          max_episode_steps=40)
    torch.manual_seed(0)
    m = AtariModel42(env.act_dim).to(dev)
    w = (m.conv1.weight, m.conv1.bias, m.conv2.weight, m.conv2.bias)
    env.reset()
    g = torch.Generator(device='cpu').manual_seed(2)
    seen = set()
    for i in range(30):
        if env.t >= env.horizon:
            env.roll()
        ref = env.current_obs_ref()
        assert isinstance(ref, ops.RingObservation) and ref.shape == (E, 4, 42, 42)
        stack = env.current_obs()
        assert torch.equal(ref.materialize(), stack)
        a, b = ops.atari42_conv12(ref, *w), ops.atari42_conv12(stack, *w)
        assert torch.equal(a, b), 'step %d' % i
        seen.update(int(x) for x in env.since[env.t + 3].unique().tolist())
        with torch.no_grad():
            assert torch.equal(m.policy(ref), m.policy(stack))
        env.step_async(torch.randint(0, env.act_dim, (E, ), generator=g).to(dev))
    assert seen == {0, 1, 2, 3}, seen            # stacks right after a reset were among them

    class StackOnly(torch.nn.Module):            # the same network behind the stack-reading interface
        def __init__(self, inner):
            super().__init__()
            self.inner = inner

        def policy_sample_into(self, obs, *a):
            assert torch.is_tensor(obs)
            return self.inner.policy_sample_into(obs, *a)

    outs = []
    for model in (m, StackOnly(m)):
        env2 = DeviceVectorEnv('BreakoutNoFrameskip-v4', E, dim=42, horizon=8, seed=5, device=dev, rom_bytes=_rom('breakout'))
        ro = DeviceRollout(env2, 8, seed=3)
        batch = ro.collect(model)
        outs.append({k: v.clone() for k, v in batch.items()})
    assert all(torch.equal(outs[0][k], outs[1][k]) for k in outs[0])


def test_conv1_84_reads_the_ring_like_the_gathered_stack(dev):
    """the 84x84 model's first layer on an ops.RingObservation against the materialised stack (bit-identical, the
    steps after resets included), and DeviceA2CRollout's batch with the ring-reading model against a stack-reading
    wrapper of it"""
    from parl_amd import ops
    from parl_amd.env import DeviceVectorEnv
    from parl_amd.models import AtariModel84
    from parl_amd.rollout import DeviceA2CRollout
    E = 21
    env = DeviceVectorEnv('BreakoutNoFrameskip-v4', E, dim=84, horizon=10, seed=6, device=dev, rom_bytes=_rom('breakout'),
                          max_episode_steps=40)
    torch.manual_seed(0)
    m = AtariModel84(env.act_dim).to(dev)
    env.reset()
    g = torch.Generator(device='cpu').manual_seed(3)
    seen = set()
    for i in range(24):
        if env.t >= env.horizon:
            env.roll()
        ref, stack = env.current_obs_ref(), env.current_obs()
        assert isinstance(ref, ops.RingObservation) and ref.shape == (E, 4, 84, 84)
        assert torch.equal(ops.atari84_conv1(ref, m.conv1.weight, m.conv1.bias),
                           ops.atari84_conv1(stack, m.conv1.weight, m.conv1.bias)), 'step %d' % i
        seen.update(int(x) for x in env.since[env.t + 3].unique().tolist())
        with torch.no_grad():
            (la, va), (lb, vb) = m.policy_and_value(ref), m.policy_and_value(stack)
            assert torch.equal(la, lb) and torch.equal(va, vb)
        env.step_async(torch.randint(0, env.act_dim, (E, ), generator=g).to(dev))
    assert seen == {0, 1, 2, 3}, seen

    class StackOnly(torch.nn.Module):
        def __init__(self, inner):
            super().__init__()
            self.inner = inner

        def policy_and_value(self, obs):
            assert torch.is_tensor(obs)
            return self.inner.policy_and_value(obs)

        def value(self, obs):
            assert torch.is_tensor(obs)
            return self.inner.value(obs)

    outs = []
    for model in (m, StackOnly(m)):
        env2 = DeviceVectorEnv('BreakoutNoFrameskip-v4', E, dim=84, horizon=6, seed=6, device=dev, rom_bytes=_rom('breakout'))
        ro = DeviceA2CRollout(env2, 6, 0.99, 1.0, seed=4)
        batch = ro.collect(model)
        outs.append({k: v.clone() for k, v in batch.items()})
    assert all(torch.equal(outs[0][k], outs[1][k]) for k in outs[0])


@pytest.mark.parametrize('game', ['pong', 'breakout'])
def test_translated_cartridge_equals_interpreter(dev, game):
    """The statically translated cartridge code (csrc/gen_cart_native.py) and the 6507 interpreter
    are the same machine: whole state blobs (RAM, TIA registers, CPU/RIOT/ALE/wrapper scalars), raw
    frame pairs, rewards and dones stay bit-identical over a long random rollout with resets."""
    from parl_amd.env import DeviceVectorEnv
    rom = _rom(game)
    E, steps = 64, 400
    mk = lambda native: DeviceVectorEnv(GAMES[game], E, dim=84, horizon=8, seed=9, device=dev, rom_bytes=rom,
                                        native=native, max_episode_steps=1500)
    a_env, b_env = mk(True), mk(False)
    assert a_env.native, 'library was built without the translated cartridge (roms/ missing at build time?)'
    assert not b_env.native
    assert torch.equal(a_env.reset(), b_env.reset())
    g = torch.Generator(device='cpu').manual_seed(1)
    for i in range(steps):
        act = torch.randint(0, a_env.act_dim, (E, ), generator=g).to(dev)
        oa, ra, da, _ = a_env.step(act)
        ob, rb, db, _ = b_env.step(act)
        # (the translated code leaves V and C alone where nothing can read them before they are defined again —
        # gen_cart_native.py, DEAD_FLAGS — so the saved processor status may differ from the interpreter's in exactly
        # those two bits: byte kOffScalars + 4 * S_P = 208 of the blob; everything else is compared bit for bit)
        sa, sb_ = a_env.states.view(E, -1).clone(), b_env.states.view(E, -1).clone()
        sa[:, 208] &= 0xbe
        sb_[:, 208] &= 0xbe
        assert torch.equal(sa, sb_), 'state blob, step %d' % i
        assert torch.equal(a_env.raw_frames, b_env.raw_frames), 'raw frames, step %d' % i
        assert torch.equal(ra, rb) and torch.equal(da, db) and torch.equal(oa, ob), 'step %d' % i
    a_env.check_faults()
    b_env.check_faults()


def test_reference_style_vector_env_on_device(dev, oracle):
    """The drop-in host API of the reference scripts on the real kernels:
    gym.make -> wrap_deepmind(dim, obs_format) -> VectorEnv(envs).reset()/step() with the
    reference's return types (lists of numpy obs / floats / bools / info dicts,
    parl/env/vector_env.py:34-63), bit-exact against the oracle's chain, and MonitorEnv statistics
    through get_wrapper_by_cls (benchmark/torch/a2c/actor.py:110-119)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'compat'))
    try:
        for m in ('gym', 'parl'):
            sys.modules.pop(m, None)
        import gym
        import parl
        from parl.env.atari_wrappers import wrap_deepmind, MonitorEnv, get_wrapper_by_cls
        from parl.env.vector_env import VectorEnv
        import parl_amd
        assert parl is parl_amd
        E = 3
        envs = [wrap_deepmind(gym.make('BreakoutNoFrameskip-v4'), dim=84, obs_format='NCHW') for _ in range(E)]
        assert envs[0].observation_space.shape == (4, 84, 84) and envs[0].action_space.n == 4
        with pytest.raises(RuntimeError):
            envs[0].reset()
        vec = VectorEnv(envs)
        id0 = vec.dev_env.env_id0
        orc = oracle.VecEnv(_rom('breakout'), 'breakout', E, 84, seed=0, env_id0=id0)
        obs = vec.reset()
        assert isinstance(obs, list) and len(obs) == E and obs[0].shape == (4, 84, 84) and obs[0].dtype == np.uint8
        assert np.array_equal(np.stack(obs), orc.reset())
        rng = np.random.default_rng(0)
        n_eps = 0
        for t in range(400):
            a = rng.integers(0, 4, E)
            obs, rew, done, info = vec.step(a)
            oo, orr, od = orc.step(a)
            assert np.array_equal(np.stack(obs), oo) and rew == [float(x) for x in orr] and done == [bool(x) for x in od]
            assert all(isinstance(r, float) for r in rew) and all(isinstance(d, bool) for d in done)
            for e in range(E):
                exp = orc.pop_episodes(e)
                got = list(get_wrapper_by_cls(envs[e], MonitorEnv).next_episode_results())
                assert got == [(float(r), int(n)) for r, n in exp]
                n_eps += len(got)
                assert ('episode' in info[e]) == bool(exp)
            if t % 50 == 49 or t == 399:  # MonitorEnv.get_total_steps: EVERY raw step, the running episode's too (:73-77)
                for e in range(E):
                    assert get_wrapper_by_cls(envs[e], MonitorEnv).get_total_steps() == orc.total_steps(e)
        assert n_eps >= 1
        mon = get_wrapper_by_cls(envs[0], MonitorEnv)
        assert mon.get_total_steps() > sum(mon.get_episode_lengths())   # an episode is running
    finally:
        sys.path.remove(os.path.join(ROOT, 'compat'))


@pytest.mark.parametrize('game,steps', [('pong', 80), ('breakout', 260)])
def test_full_size_vector_matches_oracle_on_a_subset_of_envs(dev, oracle, game, steps):
    """BASELINE configs[2] / [3] size: 1024 envs per GPU at 42x42.  The oracle cannot step 1024 envs
    in seconds, but envs are independent (state, actions, Philox stream keyed by the global env id),
    so single-env oracles for a spread of ids — first / last wave of a workgroup, workgroup and XCD
    boundaries, the last env — fed that env's actions must reproduce its observations, rewards and
    dones bit for bit, including Breakout's life-loss reset chains."""
    from parl_amd.env import DeviceVectorEnv
    rom = _rom(game)
    E, dim, seed = 1024, 42, 21
    ids = [0, 1, 3, 4, 255, 256, 511, 512, 777, 1023]
    env = DeviceVectorEnv(GAMES[game], E, dim=dim, horizon=8, seed=seed, device=dev, rom_bytes=rom)
    orcs = [oracle.VecEnv(rom, game, 1, dim, seed=seed, env_id0=i) for i in ids]
    obs = env.reset().cpu().numpy()
    for i, o in zip(ids, orcs):
        assert np.array_equal(obs[i], o.reset()[0]), 'reset env %d' % i
    g = torch.Generator().manual_seed(1)
    ndone = 0
    for t in range(steps):
        a = torch.randint(0, env.act_dim, (E, ), generator=g)
        ob, rew, done, _ = env.step(a.to(dev))
        ob, rew, done = ob.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        for i, o in zip(ids, orcs):
            oo, orr, od = o.step(a[i:i + 1].numpy())
            assert np.array_equal(ob[i], oo[0]) and rew[i] == orr[0] and bool(done[i]) == bool(od[0]), (t, i)
            ndone += int(od[0])
    env.check_faults()
    if game == 'breakout':
        assert ndone >= 3  # life losses inside the window


def test_step_with_folded_episode_stats_equals_separate_accumulation(dev):
    """step_async(ep_acc=...) (MonitorEnv statistics inside the frame_post launch) == step_async() followed by
    accumulate_episode_stats(): same frames, same statistics, on Breakout (episodes end inside the window)."""
    from parl_amd.env import DeviceVectorEnv
    E, T = 32, 400
    envs = [DeviceVectorEnv('BreakoutNoFrameskip-v4', E, dim=42, horizon=T, seed=3, device=dev) for _ in range(2)]
    accs = [torch.zeros(3, dtype=torch.float64, device=dev) for _ in range(2)]
    g = torch.Generator(device=dev).manual_seed(1)
    for e in envs:
        e.reset()
    for t in range(T):
        a = torch.randint(0, envs[0].act_dim, (E, ), device=dev, generator=g)
        envs[0].step_async(a, ep_acc=accs[0])
        envs[1].step_async(a)
        envs[1].accumulate_episode_stats(accs[1])
    assert torch.equal(envs[0].ring, envs[1].ring) and torch.equal(envs[0].since, envs[1].since)
    a0, a1 = accs[0].cpu().numpy(), accs[1].cpu().numpy()
    assert a0[0] == a1[0] and a0[0] > 0 and a0[2] == a1[2]
    np.testing.assert_allclose(a0[1], a1[1], rtol=1e-12)
    for e in envs:
        e.check_faults()


def _long_parity(dev, oracle, game, E, dim, steps, seed, ids=None, max_episode_steps=400000):
    """Whole games: the device vector and one single-env oracle per checked env id (the oracle's C code releases
    the GIL: one thread per env, running while the device steps) on the same pre-drawn random actions.  Returns per
    checked env (dones seen, MonitorEnv episodes [(step, return, length)]) after asserting obs / reward / done /
    episode records equal at every step."""
    from concurrent.futures import ThreadPoolExecutor
    from parl_amd.env import DeviceVectorEnv
    rom = _rom(game)
    ids = list(range(E)) if ids is None else list(ids)
    env = DeviceVectorEnv(GAMES[game], E, dim=dim, horizon=8, seed=seed, device=dev, rom_bytes=rom,
                          max_episode_steps=max_episode_steps)
    rng = np.random.default_rng(seed)
    acts = rng.integers(0, env.act_dim, (steps, E))
    orcs = [oracle.VecEnv(rom, game, 1, dim, seed=seed, env_id0=i, max_episode_steps=max_episode_steps) for i in ids]

    def run_oracle(k):
        o, e = orcs[k], ids[k]
        obs = np.zeros((steps + 1, 4, dim, dim), np.uint8)
        rew, done, eps = np.zeros(steps, np.float32), np.zeros(steps, np.uint8), []
        obs[0] = o.reset()[0]
        for t in range(steps):
            oo, r, d = o.step(acts[t, e:e + 1])
            obs[t + 1], rew[t], done[t] = oo[0], r[0], d[0]
            eps += [(t, ret, ln) for ret, ln in o.pop_episodes(0)]
        return obs, rew, done, eps

    sel = torch.tensor(ids, device=dev)
    with ThreadPoolExecutor(min(8, len(ids))) as ex:
        futs = [ex.submit(run_oracle, k) for k in range(len(ids))]
        d_obs = [env.reset()[sel].cpu().numpy()]
        d_rew, d_done, d_eps = [], [], [[] for _ in ids]
        for t in range(steps):
            o, r, d, info = env.step(torch.from_numpy(acts[t]).to(dev))
            d_obs.append(o[sel].cpu().numpy())
            d_rew.append(r[sel].cpu().numpy())
            d_done.append(d[sel].cpu().numpy().astype(np.uint8))
            ln, rt = info['episode_lengths'][sel].cpu().numpy(), info['episode_returns'][sel].cpu().numpy()
            for k in range(len(ids)):
                if ln[k]:
                    d_eps[k].append((t, float(rt[k]), int(ln[k])))
        res = [f.result() for f in futs]
    env.check_faults()
    out = []
    for k, (obs, rew, done, eps) in enumerate(res):
        for t in range(steps):
            assert d_rew[t][k] == rew[t], 'reward, env %d step %d' % (ids[k], t)
            assert d_done[t][k] == done[t], 'done, env %d step %d' % (ids[k], t)
            assert np.array_equal(d_obs[t + 1][k], obs[t + 1]), 'obs, env %d step %d' % (ids[k], t)
        assert np.array_equal(d_obs[0][k], obs[0])
        assert d_eps[k] == [(t, float(r), int(n)) for t, r, n in eps], 'MonitorEnv records, env %d' % ids[k]
        out.append((int(done.sum()), eps))
    return out


def test_pong_whole_games_match_oracle(dev, oracle):
    """Long horizon (the windows above are 24-260 steps): 1200 agent steps x 4 envs of Pong under random play.
    Every env plays a WHOLE game to 21 points — the cartridge's terminal -> EpisodicLifeEnv's was_real_done ->
    the real reset (ALE reset + noops, parl/env/atari_wrappers.py:103-133,177-211) -> the next game's first
    ~100+ steps — bit-exact against the oracle at every step."""
    res = _long_parity(dev, oracle, 'pong', E=4, dim=42, steps=1200, seed=3)
    for nd, eps in res:
        assert nd >= 1 and len(eps) >= 1
        t_end, ret, ln = eps[0]
        assert ret <= -15 and t_end <= 1100, (t_end, ret, ln)   # a game LOST 21 : x, not a TimeLimit cut,
        assert ln > 3000                                        # with >= 100 steps of the next game after it


def test_breakout_whole_games_match_oracle(dev, oracle):
    """700 agent steps x 4 envs of Breakout: every env sees several REAL game overs (fifth life lost ->
    was_real_done -> ALE reset + noops + FIRE) and, between them, the life-loss branch of EpisodicLifeEnv.reset
    (a NOOP step instead of a reset, atari_wrappers.py:200-211) followed by FireResetEnv's two steps (:163-171)."""
    res = _long_parity(dev, oracle, 'breakout', E=4, dim=84, steps=700, seed=3)
    for nd, eps in res:
        assert len(eps) >= 2, eps               # real game overs (MonitorEnv sits below EpisodicLifeEnv)
        assert nd >= len(eps) + 8, (nd, eps)    # dones that were life losses, not game overs


@pytest.mark.parametrize('game,E,steps', [('pong', 1024, 50), ('breakout', 256, 150)])
def test_full_size_vector_84_matches_oracle_on_a_subset_of_envs(dev, oracle, game, E, steps):
    """the 84x84 frame size at full vector width: 1024 envs (configs[2] at the north-star frame size) and
    configs[1]'s 256 envs, a spread of env ids against single-env oracles"""
    ids = [0, 1, 3, 4, 255] + ([256, 511, 512, 777, 1023] if E == 1024 else [5, 64, 127, 128, 200])
    res = _long_parity(dev, oracle, game, E=E, dim=84, steps=steps, seed=17, ids=ids)
    if game == 'breakout':
        assert sum(nd for nd, _ in res) >= 3


@pytest.mark.parametrize('game', ['pong', 'breakout'])
def test_rollout_with_head_in_the_env_launch_equals_separate_launches(dev, game, monkeypatch):
    """DeviceRollout's step as conv12 -> trunk GEMM -> ONE env launch (policy head + draw at its head, observation
    at its tail: parlhip_atari_vec_step_policy_obs) against the same rollout with the head + draw and frame_post
    as launches of their own: logits, actions, rewards, dones and observations identical, also across rollouts
    and as hipGraph segments"""
    from parl_amd.env import DeviceVectorEnv
    from parl_amd.models import AtariModel42
    from parl_amd.rollout import DeviceRollout
    rom = _rom(game)
    E, T = 12, 10
    torch.manual_seed(3)
    model = None
    outs = []
    for fused in (True, False):
        monkeypatch.setenv('PARL_AMD_FUSED_HEAD', '1' if fused else '0')
        env = DeviceVectorEnv(GAMES[game], E, dim=42, horizon=T, seed=21, device=dev, rom_bytes=rom,
                              max_episode_steps=900)
        env.fused_obs = fused
        if model is None:
            model = AtariModel42(env.act_dim).to(dev)
        ro = DeviceRollout(env, T, seed=77)
        assert ro._head_in_env_step(model) == fused
        got = []
        side = torch.cuda.Stream(device=dev)   # (a hipGraph is captured on a non-default stream)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for it in range(4):
                ro.collect_begin()
                ro.collect_segment(model, 0, T // 2, graph=True)     # eager, capture, replay, replay
                ro.collect_segment(model, T // 2, T, graph=False)
                b = ro.collect_end()
                got.append({k: (v.materialize() if hasattr(v, 'materialize') else v).clone() for k, v in b.items()})
            got.append({'ep': ro.ep_stats.clone()})
        side.synchronize()
        env.check_faults()
        outs.append(got)
    for a, b in zip(*outs):
        for k in a:
            assert torch.equal(a[k], b[k]), k
