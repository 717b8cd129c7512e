#!/usr/bin/env python
"""bench.py — IMPALA on PongNoFrameskip-v4, actors and learner on the same MI355X(s).

Metric (BASELINE.json): env frames/sec (whole job) + learner updates/sec.
One timed "step" = one full actor-learner iteration of the reference's IMPALA example
(examples/IMPALA, config examples/IMPALA/impala_config.py) executed on the device:
    T = sample_batch_steps env steps for every env (policy forward -> categorical sample ->
    emulator 4 frames -> max/gray/resize -> ring), and the learner consuming the previous T*E-row rollout
    at the REFERENCE's learner batch: E // 20 updates of train_batch_size = 1000 rows (20 sequences of
    T = 50, impala_config.py:26-31), each update (fwd, heads + fused V-trace loss, bwd, [RCCL grad
    all-reduce], global-norm clip, Adam) one hipGraph replay.  `--train-batch 0` is the other learner mode
    (ONE update per step on the whole T*E batch); it rides on the default line as the leg
    `impala_one_update`, which is also where the V-trace loss kernel sees its workload shape (T=50, B=1024).
As in the reference (actors and learner are decoupled, the behaviour policy lags the learner),
the updates on batch i-1 run on a second HIP stream WHILE batch i is collected
(parl_amd.rollout.AsyncActorLearner; --no-overlap runs them back to back).  Every timed step
contains exactly one full rollout and the full learner pass over the previous one.  Nothing is skipped
inside the timed region.  Frames counted = emulated 2600 frames of agent
steps (4 per step, frame-skip 4); reset frames are not counted.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import parl_amd as parl  # noqa: E402
from parl_amd import dist as pdist  # noqa: E402
from parl_amd import ops  # noqa: E402
from parl_amd.env import DeviceVectorEnv  # noqa: E402
from parl_amd.models import AtariModel42, AtariModel84  # noqa: E402
from parl_amd.rollout import AsyncActorLearner, DeviceRollout, ElasticDeviceRollout  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (guides/MI355X_MICROARCH.md)


class KernelTimer(object):
    """HIP-event timing of one C-ABI call, on the stream the kernel is launched on (torch's
    current stream — the one _native.stream_ptr() hands to the library)."""

    def __init__(self):
        self.pairs = []
        self.enabled = False

    def wrap(self, fn):
        def timed(*a, **k):
            if not self.enabled or torch.cuda.is_current_stream_capturing():  # (an event pair inside a capture times nothing)
                return fn(*a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = fn(*a, **k)
            e.record()
            self.pairs.append((s, e))
            return out

        return timed

    def mean_seconds(self):
        if not self.pairs:
            return None
        return sum(s.elapsed_time(e) for s, e in self.pairs) * 1e-3 / len(self.pairs)

    def stats(self):
        """{'n', 'mean', 'median', 'min', 'max'} of the launches in microseconds"""
        if not self.pairs:
            return None
        ts = sorted(s.elapsed_time(e) * 1e3 for s, e in self.pairs)
        return {'n': len(ts), 'mean': sum(ts) / len(ts), 'median': ts[len(ts) // 2], 'p10': ts[len(ts) // 10],
                'p90': ts[(len(ts) * 9) // 10], 'min': ts[0], 'max': ts[-1]}


# SQ counters of atari_env_kernel<Pong> at E = 1024 (profiles/r05_env_pmc.log: the kernel with the policy head at its head
# and the observation at its tail): active instructions per wave-clock of the two waves an env occupies
ENV_PMC = {'game': 'PongNoFrameskip-v4', 'envs': 1024, 'dim': 42,
           'issue_slot_utilisation': 69620.0 / 160725.0, 'instructions_per_frame': 69620,
           'source': 'profiles/r06_env_pmc.log (rocprofv3 --pmc, tools/pmc_env.sh): SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES per '
                     'wave and emulated frame, both in 4-clock issue slots — a wave of this kernel issues in 43 % of its '
                     'slots, two such waves share a SIMD'}


def pmc_traffic(key):
    """HBM bytes per launch from the committed rocprofv3 PMC measurement of the same kernel and
    shape (profiles/r01_scan_hbm_traffic.json, produced by tools/prof_traffic.sh: separate
    FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 read correction).  bench.py cannot run under two
    rocprofv3 passes itself; the source file is named next to the number."""
    for name in ('r06_hbm_traffic.json', 'r05_hbm_traffic.json', 'r04_hbm_traffic.json', 'r03_hbm_traffic.json', 'r02_hbm_traffic.json', 'r01e_scan_hbm_traffic.json', 'r01_scan_hbm_traffic.json'):
        try:
            t = json.load(open(os.path.join(ROOT, 'profiles', name)))[key]
            return {'traffic': t['hbm_traffic_bytes'], 'traffic_source': 'profiles/%s:%s' % (name, key)}
        except (OSError, KeyError, ValueError):
            continue
    return {'traffic': None}


def kernel_only_profile(by):
    try:
        d = json.load(open(os.path.join(ROOT, 'profiles', 'r06_heads_loss_kernel_only.json')))
        return {'us': d['avg_us'], 'frac': by / (d['avg_us'] * 1e-6) / 1e9 / HBM_PEAK_GBPS, 'source': d['source'],
                'measured_in_this_run': False}
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(game, dim, seconds_target=12.0):
    """The CPU oracle (a port, 'kind': 'port') stepping the same env chain on host cores."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import c_oracle
    from parl_amd.env import find_rom, GAMES
    name = GAMES[game][0]
    rom = find_rom(name)
    cores = max(1, min(os.cpu_count() or 1, 16))
    envs_per = 2

    def run(i):
        v = c_oracle.VecEnv(rom, name, envs_per, dim, seed=100 + i)
        v.reset()
        rng = np.random.default_rng(i)
        n, t0 = 0, time.time()
        while time.time() - t0 < seconds_target:
            for _ in range(10):
                v.step(rng.integers(0, v.num_actions, envs_per))
            n += 10
        return n * envs_per * 4, time.time() - t0

    with ThreadPoolExecutor(cores) as ex:  # ctypes releases the GIL inside the C oracle
        res = list(ex.map(run, range(cores)))
    frames = sum(r[0] for r in res)
    dt = max(r[1] for r in res)
    return {
        'value': frames / dt,
        'unit': 'env frames/s',
        'cores': cores,
        'kind': 'port',
        'sample': '%d oracle envs (%d threads x %d), random actions, %.0f s: emulator + wrapper chain + '
        'frame_post only (no policy/learner); the reference xparl+gym+ALE actor pool cannot run here' %
        (cores * envs_per, cores, envs_per, dt),
    }


def _event_time(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e-3


def heads_loss_bytes(T, B, A):
    """algorithmic HBM bytes of one impala_heads_loss_kernel launch: per (t, b) row the trunk
    output in (1024 B), its gradient out (1024 B), behaviour logits, action, reward, done in; vs, pg_adv out for
    the T-1 rows that have a successor"""
    return T * B * (2 * 256 * 4 + A * 4 + 8 + 4 + 1) + (T - 1) * B * 8


def wrap_lib(name, timer):
    """time the C-ABI entry itself (the launches on the current stream), not the Python wrapper around it
    (output allocations, the zero fill of the sums); returns an undo function"""
    from parl_amd import _native
    orig = getattr(_native.lib(), name)
    setattr(_native.lib(), name, timer.wrap(orig))
    return lambda: setattr(_native.lib(), name, orig)


def heads_loss_alone(dev, T, B, A, iters=40):
    """the V-trace loss kernel (+ its partial-sum kernel) at the workload shape on an otherwise idle GPU:
    HIP events around the C-ABI call, mean over `iters` launches after 5 warm-ups"""
    g = torch.Generator(device=dev).manual_seed(1)
    hd = torch.relu(torch.randn(T, B, 256, device=dev, generator=g))
    hw = [torch.randn(A, 256, device=dev, generator=g) * 0.1, torch.zeros(A, device=dev),
          torch.randn(1, 256, device=dev, generator=g) * 0.05, torch.zeros(1, device=dev)]
    hb = [torch.randn(T, B, A, device=dev, generator=g), torch.randint(0, A, (T, B), device=dev, generator=g),
          torch.randn(T, B, device=dev, generator=g), torch.rand(T, B, device=dev, generator=g) < 0.01]
    tm = KernelTimer()
    undo = wrap_lib('parlhip_impala_heads_loss_f32', tm)
    try:
        for i in range(iters + 5):
            tm.enabled = i >= 5
            ops.impala_heads_loss(hd, *hw, *hb, 0.99, 1.0, 1.0, 0.5, -0.01)
        torch.cuda.synchronize()
        return tm.mean_seconds(), tm.stats()
    finally:
        undo()


def heads_loss_back_to_back(dev, T, B, A, n=50):
    """the same launch pair with nothing between consecutive calls: the C-ABI entry is called `n` times in a row
    on pre-allocated buffers (no Python wrapper, no allocation, no zero fill) between ONE pair of HIP events, so
    the launch gaps of a single bracketed call drop out and what is left per call is the device time of
    impala_heads_loss_q_kernel + heads_partial_sum_kernel — measured in this run, next to the rocprof
    figure of the committed profile"""
    from parl_amd import _native
    g = torch.Generator(device=dev).manual_seed(1)
    hd = torch.relu(torch.randn(T, B, 256, device=dev, generator=g))
    hw = [torch.randn(A, 256, device=dev, generator=g) * 0.1, torch.zeros(A, device=dev),
          torch.randn(1, 256, device=dev, generator=g) * 0.05, torch.zeros(1, device=dev)]
    hb = [torch.randn(T, B, A, device=dev, generator=g), torch.randint(0, A, (T, B), device=dev, generator=g),
          torch.randn(T, B, device=dev, generator=g), torch.rand(T, B, device=dev, generator=g) < 0.01]
    lib, name, seen = _native.lib(), 'parlhip_impala_heads_loss_f32', []
    orig = getattr(lib, name)

    def record(*a):
        seen.append(a)
        return orig(*a)

    setattr(lib, name, record)
    try:
        keep = ops.impala_heads_loss(hd, *hw, *hb, 0.99, 1.0, 1.0, 0.5, -0.01)   # (its outputs stay alive: `keep`)
    finally:
        setattr(lib, name, orig)
    if not seen or keep is None:
        return None
    args = seen[0]
    for _ in range(5):
        orig(*args)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        orig(*args)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / n


def one_update_leg(dev, E, T, dim, game, K, learn_rows, env_id0=0, warm=2):
    """The pipeline in its other learner mode — ONE update per step on the whole T*E-row rollout (round 1-3's
    headline) — K timed steps.  This is where impala_heads_loss_kernel runs at the workload shape
    (T=50, B=1024, A=6: 107 MB per launch) BESIDE the actors; every launch of the timed steps is
    bracketed by HIP events on the learner stream."""
    env = DeviceVectorEnv(game, E, dim=dim, horizon=T, seed=1234, env_id0=env_id0, device=dev)
    model = (AtariModel42 if dim == 42 else AtariModel84)(env.act_dim).to(dev)
    alg = parl.algorithms.IMPALA(model, sample_batch_steps=T, gamma=0.99, vf_loss_coeff=0.5,
                                 clip_rho_threshold=1.0, clip_pg_rho_threshold=1.0)
    alg.max_learn_rows = learn_rows or None
    pipe = AsyncActorLearner(alg, [env], T, seed=99)
    lr_s = parl.utils.PiecewiseScheduler([(0, 0.001), (20000, 0.0005), (40000, 0.0001)])
    tm = KernelTimer()
    undo = wrap_lib('parlhip_impala_heads_loss_f32', tm)
    try:
        pipe.prime()
        for _ in range(warm):
            pipe.step(lr_s.step(), -0.01)
        pipe.synchronize()
        torch.cuda.synchronize()
        tm.enabled = True
        t0 = time.time()
        for _ in range(K):
            loss, kl = pipe.step(lr_s.step(), -0.01)
        pipe.synchronize()
        torch.cuda.synchronize()
        dt = time.time() - t0
    finally:
        undo()
    assert np.isfinite(float(loss.total_loss))
    env.check_faults()
    return {'workload': 'BASELINE configs[2] with ONE learner update per step on the whole %d-row rollout (%d-row passes, '
                        'one V-trace loss launch): %s IMPALA, %d actors, T=%d, %dx%d, actor/learner overlapped' %
                        (T * E, learn_rows or T * E, game, E, T, dim, dim),
            'env_frames_per_s': K * T * E * 4 / dt, 'updates_per_s': K / dt, 'ms_per_step': dt / K * 1e3, 'steps': K,
            'heads_loss_in_pipeline_s': tm.mean_seconds(), 'heads_loss_in_pipeline_stats_us': tm.stats(),
            'act_dim': env.act_dim}


def torch_cpu_baselines(threads=8):
    """BASELINE.md section 3's remaining CPU baselines (measurement infrastructure, after the timed regions):
    the learner update on torch-CPU with the reference's own parl/algorithms/torch/a2c.py + ActorCritic
    (staged byte for byte into oracle/_ref/ by build(): kind "reference"), and the per-t V-trace loop of
    vtrace.py:99-137 as torch-CPU ops (kind "port"), both at `threads` threads."""
    from oracle import ref_torch_baselines
    return ref_torch_baselines.time_all(threads=min(threads, os.cpu_count() or 1))


def extra_legs(dev, only=None):
    """Short runs of the other BASELINE.json configs on ONE GPU (each leg: its own envs / model,
    1 warm-up + a few timed iterations, wall-clock with synchronize on both sides; frames = emulated
    2600 frames of agent steps).  They ride on the headline's JSON line under their own keys and
    never touch `value`."""
    from parl_amd.rollout import DeviceA2CRollout
    out = {}

    def guarded(name, fn):
        if only and name not in only:
            return
        try:
            out[name] = fn()
        except Exception as e:  # a leg must never take the headline down
            out[name] = {'error': '%s: %s' % (type(e).__name__, e)}
        torch.cuda.synchronize()
        torch.cuda.empty_cache()

    # ---- configs[1]: Pong A2C, 256 on-GPU envs, 84x84, T=20, lambda=1 (examples/A2C/a2c_config.py) ----
    def a2c_c2():
        E, T, K = 256, 20, 30
        env = DeviceVectorEnv('PongNoFrameskip-v4', E, dim=84, horizon=T, seed=7, device=dev)
        model = AtariModel84(env.act_dim).to(dev)
        alg = parl.algorithms.A2C(model, vf_loss_coeff=0.5)
        ro = DeviceA2CRollout(env, T, gamma=0.99, lam=1.0, seed=3)
        lr_s = parl.utils.LinearDecayScheduler(0.001, int(1e7))

        def step():
            b = ro.collect(model)
            return alg.learn(b['obs'], b['actions'], b['advantages'], b['target_values'], lr_s.step(T * E), -0.01)

        step()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(K):
            loss = step()
        torch.cuda.synchronize()
        dt = time.time() - t0
        assert np.isfinite(float(loss[0]))
        env.check_faults()
        rew, val = torch.randn((T, E), device=dev), torch.randn((T, E), device=dev)
        dn = torch.rand((T, E), device=dev) < 1 / 800
        nv = torch.randn(E, device=dev)
        g = _event_time(lambda: ops.gae(rew, val, dn, nv, 0.99, 1.0))
        by = T * E * 17
        return {'workload': 'BASELINE configs[1]: PongNoFrameskip-v4 A2C, 256 on-GPU envs, 84x84, T=20, lambda=1.0; '
                            'rollout then update (synchronous A2C), convolutions on the MFMA kernels',
                'rollout_then_update_is_serial_by_contract': 'examples/A2C/train.py:82-94 pushes the new weights to every '
                'actor before each sample and waits for all of them before the one learn(): the emulator idles during '
                'the update (~4.5 of ~21 ms per iteration) by the algorithm, not by an omission of overlap',
                'env_frames_per_s': K * T * E * 4 / dt, 'updates_per_s': K / dt, 'ms_per_step': dt / K * 1e3, 'steps': K,
                'gae_kernel': {'shape': 'T=20 B=256 u8 dones', 'us': g * 1e6, 'bytes': by, 'GBps': by / g / 1e9,
                               'frac_of_hbm_peak': by / g / 1e9 / HBM_PEAK_GBPS,
                               'note': '87 KB: launch-latency-bound by construction (SURVEY 8d)'}}

    # ---- IMPALA at 84x84 (the north-star frame size), 1024 envs, T=50 ----
    def impala_84():
        E, T, K = 1024, 50, 10
        env = DeviceVectorEnv('PongNoFrameskip-v4', E, dim=84, horizon=T, seed=8, device=dev)
        model = AtariModel84(env.act_dim).to(dev)
        alg = parl.algorithms.IMPALA(model, sample_batch_steps=T, gamma=0.99, vf_loss_coeff=0.5,
                                     clip_rho_threshold=1.0, clip_pg_rho_threshold=1.0)
        # the update in 8 chunks of 128 sequences: the backward kernels are persistent (one workgroup per CU
        # until the chunk is done) and share no CU with the actors' conv kernels (LDS), so one 51,200-row
        # pass stalls the rollout for its whole length (measured: 1.26 M frames/s in one pass, 1.58 M in 8)
        alg.max_learn_rows = 6400
        pipe = AsyncActorLearner(alg, [env], T, seed=4)
        pipe.prime()
        pipe.step(0.001, -0.01)
        pipe.synchronize()
        t0 = time.time()
        for _ in range(K):
            loss, kl = pipe.step(0.001, -0.01)
        pipe.synchronize()
        torch.cuda.synchronize()
        dt = time.time() - t0
        assert np.isfinite(float(loss.total_loss))
        env.check_faults()
        res = {'workload': 'PongNoFrameskip-v4 IMPALA V-trace at 84x84 (the north-star frame size), 1024 actors, T=50, '
                           'actor/learner overlapped; ONE update per step on the 51,200-row batch (8 chunks accumulated); convolutions of actors '
                           'and learner on the MFMA kernels (conv1_84 / conv23_84 forward, three backward kernels)',
               'env_frames_per_s': K * T * E * 4 / dt, 'updates_per_s': K / dt, 'ms_per_step': dt / K * 1e3, 'steps': K}
        # the same with the reference's 1000-row learner updates (hipGraph replays, mid-rollout weight refresh)
        del pipe, alg, model, env
        torch.cuda.empty_cache()
        env = DeviceVectorEnv('PongNoFrameskip-v4', E, dim=84, horizon=T, seed=8, device=dev)
        model = AtariModel84(env.act_dim).to(dev)
        alg = parl.algorithms.IMPALA(model, sample_batch_steps=T, gamma=0.99, vf_loss_coeff=0.5,
                                     clip_rho_threshold=1.0, clip_pg_rho_threshold=1.0)
        pipe = AsyncActorLearner(alg, [env], T, seed=4, train_batch_size=1000)
        pipe.prime()
        for _ in range(2):  # warm-up
            pipe.step(0.001, -0.01)
        pipe.synchronize()
        u0, t0 = pipe.updates, time.time()
        for _ in range(K):
            loss, kl = pipe.step(0.001, -0.01)
        pipe.synchronize()
        torch.cuda.synchronize()
        dt = time.time() - t0
        assert np.isfinite(float(loss.total_loss))
        env.check_faults()
        res['train_batch_1000'] = {'env_frames_per_s': K * T * E * 4 / dt, 'updates_per_s': (pipe.updates - u0) / dt,
                                   'ms_per_step': dt / K * 1e3, 'steps': K,
                                   'actor_weight_refresh_points': [list(x) for x in pipe.refresh_points]}
        return res

    # ---- configs[3] per GPU: Breakout IMPALA, 1024 of the 8192 actors, A=4 ----
    def breakout_c4():
        E, T, K = 1024, 50, 10
        # elastic launches: an env inside a life-loss reset drops out of the next launches instead of making
        # every launch 16 frames long (ElasticDeviceRollout); the env's horizon bounds the launches of a batch
        env = DeviceVectorEnv('BreakoutNoFrameskip-v4', E, dim=42, horizon=4 * T + 32, seed=9, device=dev)
        model = AtariModel42(env.act_dim).to(dev)
        alg = parl.algorithms.IMPALA(model, sample_batch_steps=T, gamma=0.99, vf_loss_coeff=0.5,
                                     clip_rho_threshold=1.0, clip_pg_rho_threshold=1.0)
        alg.max_learn_rows = 6400
        pipe = AsyncActorLearner(alg, [env], T, seed=5, elastic=True)
        pipe.prime()
        pipe.step(0.001, -0.01)
        pipe.synchronize()
        t0 = time.time()
        launches = 0
        for _ in range(K):
            loss, kl = pipe.step(0.001, -0.01)
            launches += pipe.rollout.launches
        pipe.synchronize()
        torch.cuda.synchronize()
        dt = time.time() - t0
        assert np.isfinite(float(loss.total_loss))
        env.check_faults()
        return {'workload': 'BASELINE configs[3], one GPU\'s share: BreakoutNoFrameskip-v4 IMPALA, 1024 of 8192 actors, '
                            'A=4, 42x42, T=50, actor/learner overlapped, elastic launches (<= 4 frames per env and launch; '
                            'frames counted = 4 per agent step, the reset sequences\' own frames are not counted)',
                'launches_per_batch': launches / K,
                'env_frames_per_s': K * T * E * 4 / dt, 'updates_per_s': K / dt, 'ms_per_step': dt / K * 1e3, 'steps': K}

    # ---- configs[4]: the PPO scans at HalfCheetah shapes (T=2048, E=4096; minibatch 262,144) ----
    def ppo_c5_scans():
        T, E = 2048, 4096
        rew, val = torch.randn((T, E), device=dev).clamp_(-10, 10), torch.randn((T, E), device=dev)
        dn = (torch.rand((T, E), device=dev) < 1e-3).float()
        nv, ld = torch.randn(E, device=dev), torch.zeros(E, device=dev)
        g = _event_time(lambda: ops.gae(rew, val, dn, nv, 0.99, 0.95, last_done=ld, done_convention=1))
        by = T * E * 20
        adv = torch.randn(T * E, device=dev)
        idx = torch.randperm(T * E, device=dev)[:262144]
        a = _event_time(lambda: ops.adv_normalize(adv, idx))
        aby = 262144 * (8 + 4 + 4 + 4)  # index read, gathered read (twice), write
        return {'workload': 'BASELINE configs[4] shapes: RolloutStorage.compute_returns at T=2048 x E=4096 (f32 dones, '
                            '20 B/elt) and the per-minibatch advantage normalisation on 262,144 gathered elements',
                'gae_kernel': {'kernel': 'gae_lookback_kernel (single pass, 32-step chunks)', 'us': g * 1e6, 'bytes': by,
                               'GBps': by / g / 1e9, 'frac_of_hbm_peak': by / g / 1e9 / HBM_PEAK_GBPS,
                               'traffic_key': 'profiles/r01e_scan_hbm_traffic.json:gae_T2048_B4096_f32'},
                'adv_normalize_kernel': {'us': a * 1e6, 'bytes': aby, 'GBps': aby / a / 1e9}}

    # ---- configs[4] END TO END: the examples/PPO/train.py loop at HalfCheetah shapes ----
    def ppo_c5():
        ex = os.path.join(ROOT, 'examples', 'PPO')
        sys.path.insert(0, ex)
        try:
            from agent import PPOAgent
            from env_utils import ParallelEnv
            from mujoco_config import mujoco_config
            from mujoco_model import MujocoModel
            from parl_amd.storage import RolloutStorage
        finally:
            sys.path.remove(ex)
        from parl_amd.algorithms import PPO
        cfg = dict(mujoco_config, env_num=4096, seed=0)
        cfg['batch_size'] = cfg['env_num'] * cfg['step_nums']
        cfg['num_updates'] = 100
        envs = ParallelEnv(cfg, device=dev)
        model = MujocoModel(envs.obs_space, envs.act_space)
        ppo = PPO(model, clip_param=cfg['clip_param'], entropy_coef=cfg['entropy_coef'], initial_lr=cfg['initial_lr'],
                  continuous_action=True)
        agent = PPOAgent(ppo, cfg)
        rollout = RolloutStorage(cfg['step_nums'], cfg['env_num'], envs.obs_space, envs.act_space, device=dev)
        obs = envs.reset()
        done = torch.zeros(cfg['env_num'], device=dev)
        phases = {}

        def iteration():
            nonlocal obs, done
            torch.cuda.synchronize()
            t0 = time.time()
            for step in range(cfg['step_nums']):  # examples/PPO/train.py:90-103
                value, action, logprob, _ = agent.sample(obs)
                next_obs, reward, next_done = envs.step(action)
                rollout.append(obs, action, logprob, reward, done, value.flatten())
                obs, done = next_obs, next_done
            torch.cuda.synchronize()
            t1 = time.time()
            rollout.compute_returns(agent.value(obs).flatten(), done)  # :105-107
            torch.cuda.synchronize()
            t2 = time.time()
            out = agent.learn(rollout)  # :108: update_epochs x num_minibatches PPO.learn calls
            torch.cuda.synchronize()
            t3 = time.time()
            phases.update(rollout_s=t1 - t0, compute_returns_s=t2 - t1, learn_s=t3 - t2)
            return out, t3 - t0

        # warm-up: a short rollout's worth of steps and one minibatch pass would leave the storage half filled;
        # one full untimed iteration instead
        iteration()
        (vl, al, el, lr), dt = iteration()
        assert np.isfinite(vl) and np.isfinite(al)
        n = cfg['batch_size']
        # CPU baseline of the wrapper + returns part on a bounded sample: the C oracle's VecNormalize (one
        # RunningMeanStd per env, mujoco_wrappers.py:73-206) for 64 steps of 4096 envs, the numpy port of
        # RolloutStorage.compute_returns (storage.py:45-64) at full size
        from oracle import c_oracle, py_baselines
        vn = c_oracle.VecNormalize(4096, 17)
        rng = np.random.default_rng(0)
        raw, rw, dn = rng.standard_normal((4096, 17)), rng.standard_normal(4096), np.zeros(4096, np.uint8)
        t0 = time.time()
        for _ in range(64):
            vn.filter_obs(raw)
            vn.filter_reward(rw, dn)
        vn_s = (time.time() - t0) / 64
        T_, E_ = cfg['step_nums'], cfg['env_num']
        r_, v_ = rng.standard_normal((T_, E_)).astype(np.float32), rng.standard_normal((T_, E_)).astype(np.float32)
        d_ = (rng.random((T_, E_)) < 1e-3).astype(np.float32)
        t0 = time.time()
        py_baselines.compute_returns(r_, v_, d_, v_[0], d_[0])
        cr_s = time.time() - t0
        return {'workload': 'BASELINE configs[4] end to end: the examples/PPO/train.py loop (train.py:90-112) with 4096 '
                            'host-stepped simulators (a synthetic stand-in with HalfCheetah shapes: MuJoCo is not in the '
                            'image), per step: policy sample on the device, actions D2H, host step, one H2D of raw f64 '
                            'obs / rewards / dones, VecNormalize + RolloutStorage.append kernels; then compute_returns '
                            '(T=2048 x E=4096) and update_epochs=10 x num_minibatches=32 PPO.learn calls on 262,144-row '
                            'minibatches gathered on the device',
                'agent_steps_per_s': n / dt, 'seconds_per_iteration': dt, 'phases': phases,
                'updates_per_s': cfg['update_epochs'] * cfg['num_minibatches'] / dt,
                'cpu_port_baseline': {'vecnormalize_step_ms_4096_envs (C oracle, 1 core)': vn_s * 1e3,
                                      'compute_returns_s_T2048_E4096 (numpy port, 1 core)': cr_s,
                                      'kind': 'port', 'cores': 1}}

    guarded('a2c_c2', a2c_c2)
    guarded('impala_84', impala_84)
    guarded('breakout_c4_per_gpu', breakout_c4)
    guarded('ppo_c5_scans', ppo_c5_scans)
    guarded('ppo_c5', ppo_c5)
    return out


def self_launch(args):
    """`python bench.py --gpus N` without torchrun: start the N ranks ourselves, exactly as the
    driver would (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py ...`).  Rank 0 prints the one JSON line.  With fewer
    visible GPUs than ranks the ranks share devices and talk over gloo (RCCL cannot put two ranks
    on one device); the JSON line says so."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    # RCCL / device-tensor sharing across the ranks' processes: this pool's host driver supports dmabuf IPC only;
    # with the legacy mode hipIpcGetMemHandle fails ("invalid argument") and the first collective with it.  The
    # image exports it already — set here too so that a scrubbed environment launches the same way (the tests'
    # own subprocesses set it for the same reason).
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if torch.cuda.device_count() < args.gpus:
        env['PARL_AMD_SHARE_GPU'] = '1'
        env['PARL_AMD_DIST_BACKEND'] = 'gloo'
        # N processes on one device: every stream priority level of every process wants its own hardware queue;
        # let the runtime multiplex streams onto two queues per process instead of exhausting the device's
        env.setdefault('GPU_MAX_HW_QUEUES', '2')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


class Watchdog(object):
    """A rank that stops making progress (a collective a peer never joins, a wedged GPU) must not hang silently:
    a daemon thread prints ONE JSON error line and ends the process when `beat()` was not called for `limit`
    seconds.  (RCCL's own watchdog aborts the process after the process-group timeout; this one also covers hangs
    outside collectives and leaves a line a driver can parse.)"""

    def __init__(self, limit):
        import threading
        self.limit, self.phase, self.last, self.rank = float(limit), 'start', time.time(), int(os.environ.get('RANK', '0'))
        if self.limit > 0:
            threading.Thread(target=self._run, daemon=True).start()

    def beat(self, phase=None):
        self.last = time.time()
        if phase is not None:
            self.phase = phase

    def _run(self):
        while True:
            time.sleep(1.0)
            idle = time.time() - self.last
            if idle > self.limit:
                print(json.dumps({'error': 'no progress for %.0f s in phase %r' % (idle, self.phase), 'rank': self.rank,
                                  'world_size_env': int(os.environ.get('WORLD_SIZE', '1')),
                                  'collectives': pdist.describe(), 'rccl_log_tail': pdist.collective_log_tail()}),
                      flush=True)
                os._exit(3)


def main():
    try:
        _main()
    except SystemExit:
        raise
    except BaseException as e:  # every rank leaves ONE parseable line and a non-zero exit code
        import traceback
        traceback.print_exc()
        print(json.dumps({'error': '%s: %s' % (type(e).__name__, e), 'rank': int(os.environ.get('RANK', '0')),
                          'world_size_env': int(os.environ.get('WORLD_SIZE', '1')),
                          'collectives': pdist.describe(), 'rccl_log_tail': pdist.collective_log_tail()}), flush=True)
        sys.exit(2)


def _main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)   # (rollout segments are captured as hipGraphs on their second run: fewer warm-up steps put captures into the timed region)
    ap.add_argument('--envs', type=int, default=1024, help='envs (actors) per GPU')
    ap.add_argument('--dim', type=int, default=42, help='obs size: 42 = examples/IMPALA config, 84 = A2C model')
    ap.add_argument('--sample-batch-steps', type=int, default=50)
    ap.add_argument('--game', default='PongNoFrameskip-v4')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--actor-groups', type=int, default=1, help='env groups (actor streams) per GPU')
    ap.add_argument('--learn-rows', type=int, default=6400,
                    help='rows per network forward / backward pass of the ONE learner update per step '
                    '(IMPALA.max_learn_rows, chunk mode "forward": the V-trace loss kernel still runs once on the '
                    'whole batch; 0: the whole batch in one pass).  The backward kernels are persistent and '
                    'share no CU with the actors\' conv kernels, so shorter passes stall the rollout less: '
                    '2.54 M frames/s in one pass, 2.61-2.64 M at 6400 rows')
    ap.add_argument('--train-batch', type=int, default=1000,
                    help='rows per learner update and rank.  1000 (default) = the reference\'s train_batch_size '
                    '(impala_config.py:31): the rollout is consumed as E // 20 updates of 20 sequences, each one '
                    'hipGraph replay (GraphedLearn).  0: ONE update per step on the whole T*E rollout')
    ap.add_argument('--elastic', choices=('auto', 'on', 'off'), default='auto',
                    help='elastic launches (ElasticDeviceRollout): auto = games with lives (Breakout)')
    ap.add_argument('--quick', action='store_true',
                    help='headline workload only: skip the saturating-shape roofline and the extra config legs')
    ap.add_argument('--no-overlap', action='store_true',
                    help='run rollout and learner update back to back on one stream instead of overlapped')
    ap.add_argument('--hang-timeout', type=float, default=float(os.environ.get('PARL_AMD_HANG_TIMEOUT', '240')),
                    help='seconds without progress after which a rank prints a JSON error line and exits 3 (0: off)')
    ap.add_argument('--only-legs', default='',
                    help='dev: run only these extra legs (comma separated) and print their JSON, no headline run')
    args = ap.parse_args()

    if args.only_legs:
        dev = torch.device('cuda', 0)
        torch.cuda.set_device(dev)
        print(json.dumps(extra_legs(dev, only=set(args.only_legs.split(',')))))
        return
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return self_launch(args)
    wd = Watchdog(args.hang_timeout)
    wd.beat('rendezvous')
    rank, local, world = pdist.init()
    wd.beat('setup')
    if world != args.gpus:
        sys.exit('bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU, or run without torchrun and '
                 'let bench.py start the ranks itself)' % (args.gpus, world))
    shared = bool(os.environ.get('PARL_AMD_SHARE_GPU'))
    if shared:  # fewer GPUs than ranks (a 1-GPU test box): ranks share devices, gloo instead of RCCL
        local = local % torch.cuda.device_count()
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    torch.manual_seed(0)
    E, T, dim = args.envs, args.sample_batch_steps, args.dim

    # reference config: examples/IMPALA/impala_config.py:15-46
    cfg = dict(gamma=0.99, vf_loss_coeff=0.5, clip_rho_threshold=1.0, clip_pg_rho_threshold=1.0,
               lr_scheduler=[(0, 0.001), (20000, 0.0005), (40000, 0.0001)], entropy_coeff_scheduler=[(0, -0.01)])
    # env groups: the E envs of this GPU are stepped as G independent groups on G streams so that one
    # group's policy forward overlaps the other groups' emulator kernels (env ids / RNG streams are
    # those of a single E-env vector)
    G = 1 if args.no_overlap else max(1, args.actor_groups)
    assert E % G == 0
    Eg = E // G
    elastic = args.elastic == 'on' or (args.elastic == 'auto' and 'Breakout' in args.game and G == 1)
    envs = [DeviceVectorEnv(args.game, Eg, dim=dim, horizon=4 * T + 32 if elastic else T, seed=1234,
                            env_id0=rank * E + g * Eg, device=dev)
            for g in range(G)]
    env = envs[0]
    model = (AtariModel42 if dim == 42 else AtariModel84)(env.act_dim).to(dev)
    alg = parl.algorithms.IMPALA(model, sample_batch_steps=T, gamma=cfg['gamma'], vf_loss_coeff=cfg['vf_loss_coeff'],
                                 clip_rho_threshold=cfg['clip_rho_threshold'],
                                 clip_pg_rho_threshold=cfg['clip_pg_rho_threshold'])
    alg.max_learn_rows = args.learn_rows or None
    pdist.broadcast_model(model)
    if pdist.active():  # also a one-rank group (PARL_AMD_FORCE_DIST=1): the whole DP path over RCCL
        alg.grad_hook = pdist.FlatGradAllReduce(model)
    lr_s = parl.utils.PiecewiseScheduler(cfg['lr_scheduler'])
    ent_s = parl.utils.PiecewiseScheduler(cfg['entropy_coeff_scheduler'])

    vt_timer, hl_timer = KernelTimer(), KernelTimer()
    for name in ('parlhip_impala_loss_f32', 'parlhip_vtrace_from_logits_f32'):
        wrap_lib(name, vt_timer)
    undo_hl = wrap_lib('parlhip_impala_heads_loss_f32', hl_timer)  # host calls only: --train-batch 0 / --no-overlap
    env_timer, fp_timer = KernelTimer(), KernelTimer()
    for e in envs:
        e.step_async = env_timer.wrap(e.step_async)
        e.step_policy_async = env_timer.wrap(e.step_policy_async)
        e.step_elastic_async = env_timer.wrap(e.step_elastic_async)
        e._frame_post = fp_timer.wrap(e._frame_post)
        e._frame_post_elastic = fp_timer.wrap(e._frame_post_elastic)

    if args.no_overlap:
        pipe = None
        rollout = (ElasticDeviceRollout if elastic else DeviceRollout)(env, T, seed=99)

        def step():
            batch = rollout.collect(model)
            loss, kl = alg.learn(batch['obs'], batch['actions'], batch['behaviour_logits'], batch['rewards'],
                                 batch['dones'], lr_s.step(), ent_s.step(), time_major=True)
            if pdist.active():  # small-tensor trajectory all-gather (global statistics), SURVEY 8e
                pdist.all_gather_small({'rewards': rollout.rewards, 'dones': rollout.dones,
                                        'actions': rollout.actions})
            return loss
    else:
        # IMPALA's actor/learner decoupling on one GPU: the learner update on batch i-1 runs on its
        # own stream while the actors collect batch i (behaviour policy lags by one update)
        tb = args.train_batch if (args.train_batch and G == 1 and T * E > args.train_batch) else 0
        pipe = AsyncActorLearner(alg, envs, T, seed=99, elastic=elastic, train_batch_size=tb or None)
        rollout = pipe.rollout
        pipe.prime()  # untimed: every timed step = one rollout + one learner update

        pipe.gather_small = pdist.active()  # small-tensor trajectory all-gather on the learner stream

        def step():
            if pipe.sub_batches:  # the schedulers step once per update, as the reference's learner does
                loss, kl = pipe.step(lr_s, ent_s)
            else:
                loss, kl = pipe.step(lr_s.step(), ent_s.step())
            return loss

    for _ in range(args.warmup):
        step()
        wd.beat('warm-up')
    pdist.barrier()
    torch.cuda.synchronize()
    wd.beat('timed steps')
    vt_timer.enabled = hl_timer.enabled = env_timer.enabled = fp_timer.enabled = True
    updates0 = pipe.updates if pipe is not None else 0
    t0 = time.time()
    for _ in range(args.steps):
        loss = step()
        wd.beat()
    pdist.barrier()
    torch.cuda.synchronize()
    dt_local = time.time() - t0
    dt = pdist.all_reduce_max_scalar(dt_local)
    dt_ranks = pdist.all_gather_scalar(dt_local)
    wd.beat('after the timed steps')
    # the gradient all-reduce by itself on the real bucket (every rank, same point of the program; gloo on shared
    # GPUs is host-staged and only functional): what one update's exchange costs, for reading a scaling record
    allreduce_us = None
    if pdist.active() and alg.grad_hook is not None:
        if pipe is not None:
            pipe.synchronize()
        allreduce_us = pdist.time_allreduce(alg.grad_hook, pipe.learn_stream if pipe is not None else None,
                                            iters=30 if torch.distributed.get_backend() == 'nccl' else 5)
        pdist.barrier()
        wd.beat('after the all-reduce timing')
    wd.limit = max(wd.limit, 1200.0)  # the legs behind the headline (CPU baselines, profiles) have no collectives
    for e in envs:
        e.check_faults()
    total_loss = float(loss.total_loss.item())
    assert np.isfinite(total_loss)

    K = args.steps
    frames = K * T * E * 4 * world
    graphed_mode = bool(pipe is not None and pipe.sub_batches)
    n_upd = len(pipe.sub_batches) if graphed_mode else 1
    rows_upd = (pipe.sub_batches[0][1] * T) if graphed_mode else T * E
    A = env.act_dim
    out = {
        'metric': 'env frames/sec (whole job), IMPALA PongNoFrameskip-v4 actor-learner on device',
        'value': frames / dt,
        'unit': 'env frames/s',
        'n_gpus': world,
        'steps': K,
        'warmup': args.warmup,
        'ms_per_step': dt / K * 1e3,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'u8 emulation / f32 learner+scans',
        'data': 'synthetic (on-device emulation of the Pong cartridge, random-init policy)',
        'config': {
            'workload': 'BASELINE configs[2]: PongNoFrameskip-v4 IMPALA V-trace, %d actors per GPU, learner at the '
                        'reference\'s train_batch_size (impala_config.py:31)' % E if graphed_mode else
                        'BASELINE configs[2]: PongNoFrameskip-v4 IMPALA V-trace, %d actors per GPU, one learner update '
                        'per rollout' % E,
            'envs_per_gpu': E, 'sample_batch_steps': T, 'obs_dim': dim,
            'learner_mode': ('train_batch_size: %d hipGraph updates per rollout' % n_upd) if graphed_mode else
                            'one update per rollout',
            'train_batch_rows_per_update_per_rank': rows_upd,
            # data-parallel semantics: every rank contributes its own `train_batch_size` rows to an
            # update and the gradients are SUMMED (the losses are sums over rows, impala.py:67-79): one update
            # = the reference's update on the union of world x 1000 rows
            'train_batch': rows_upd * world,
            'frame_skip': 4, 'parallelism': 'dp%d (envs sharded by rank, grad all-reduce)' % world,
            'actor_learner_overlap': not args.no_overlap, 'actor_groups': G, 'elastic_launches': elastic,
            'learner_rows_per_pass': rows_upd if graphed_mode else (args.learn_rows or T * E),
            'learner_updates_per_step': n_upd,
            'actor_weight_refresh_points': [list(x) for x in pipe.refresh_points] if pipe is not None else [],
            'actor_weight_refresh_points_are': 'fixed (parl_amd.rollout.fixed_refresh_points: a function of T, updates '
                                               'per rollout and frame size; nothing calibrated at run time)',
            'env_ids_per_rank': [[r * E, r * E + E - 1] for r in range(world)],
            'process_group': pdist.describe(),
            'dp_update_form': (None if not (graphed_mode and pdist.active()) else
                               ('ONE hipGraph per update with the RCCL all-reduce captured inside (forward + backward | '
                                'all-reduce | clip + Adam)'
                                if all(g.allreduce_in_graph for g in pipe.graphed.values()) else
                                'two hipGraphs per update with an eager all-reduce between them' +
                                ''.join(' [capture of the collective failed: %s]' % g.allreduce_capture_error
                                        for g in list(pipe.graphed.values())[:1] if g.allreduce_capture_error))),
            'collectives': (('none (single process)' if not pdist.active() else 'RCCL, one-rank group (PARL_AMD_FORCE_DIST)')
                            if world == 1 else
                            ('gloo, ranks SHARE GPUs (fewer devices than ranks: functional run, not a scaling number)'
                             if shared else 'RCCL: flat-gradient all-reduce + small-tensor all-gather per update')),
        },
        'timed_region_s': dt,
        # per rank (index = rank): its own clock around the same K steps and the frames of ITS envs over it; `value`
        # uses the slowest rank's clock
        'per_rank': {'timed_region_s': dt_ranks, 'env_frames_per_s': [K * T * E * 4 / x for x in dt_ranks]},
        'grad_allreduce_alone': allreduce_us,
        'learner_updates_per_sec': ((pipe.updates - updates0) if pipe is not None else K) / dt,
        'agent_steps_per_sec': K * T * E * world / dt,
    }
    if pipe is not None and env_timer.mean_seconds() is None and not elastic:
        # the timed region replayed the actors' steps as hipGraph segments (no host call to bracket): the per-kernel
        # figures below come from ONE eager rollout of the same actors after it, alone on the device
        pipe.synchronize()
        with torch.cuda.stream(pipe.actor_stream):
            pipe.rollout.collect_begin()
            pipe.rollout.collect_steps(pipe.actor_model)
            pipe.rollout.collect_end()
        pipe.synchronize()
    es, fps = env_timer.mean_seconds(), fp_timer.mean_seconds()
    fused_obs = bool(getattr(env, 'fused_obs', False)) and not elastic
    if fps is None or fused_obs:
        # the step is ONE launch (the observation is made at the tail of the env kernel): the stand-alone
        # frame_post kernel (reset(), other frame sizes, elastic launches) is timed by itself for its roofline row
        if pipe is not None:
            pipe.synchronize()
        fps = _event_time(lambda: env._frame_post(3), iters=30)
    hl_in, hl_in_stats = hl_timer.mean_seconds(), hl_timer.stats()
    if graphed_mode:
        stats, n = pipe.pop_learn_stats()
        out['mean_losses_total_pi_vf_entropy_kl'] = stats
        # one update alone on the device (the graph replay, inputs loaded): what the learner costs the GPU
        gl = pipe.graphed[pipe.sub_batches[0][1]]
        if not pdist.active():
            out['update_alone_ms'] = _event_time(lambda: gl.replay(1e-4), iters=30) * 1e3
    undo_hl()
    one = None
    if graphed_mode and not shared and not elastic and not args.no_overlap:
        # The updates of the headline are graph replays of a T x 20-sequence batch (no host call to time, and
        # 2.1 MB per launch: launch-bound by construction).  The V-trace loss kernel meets its WORKLOAD shape
        # (T=50, B=1024: 107 MB) in the pipeline's other learner mode: K steps of it, on every rank for itself
        # (no collectives), the kernel bracketed by HIP events on the learner stream.
        del pipe, rollout, envs, env, model, alg, step
        torch.cuda.empty_cache()
        # (>= 100 in-pipeline launches of the V-trace loss kernel on the full line: its median / p10 / p90 are what
        # `roofline.frac` is computed from; --quick and N > 1 keep the leg short)
        one = one_update_leg(dev, E, T, dim, args.game, 100 if (world == 1 and not args.quick) else 5, args.learn_rows,
                             env_id0=rank * E)
        hl_in, hl_in_stats = one.pop('heads_loss_in_pipeline_s'), one.pop('heads_loss_in_pipeline_stats_us')
        torch.cuda.empty_cache()
    pdist.barrier()
    if rank == 0:
        if one is not None:
            out['impala_one_update'] = one
        # --- roofline of the V-trace kernel at the workload shape (HBM-bound scan) ---
        Bl = E if G == 1 else Eg
        if hl_in is not None or (graphed_mode and A in (4, 6) and T <= 64):
            by = heads_loss_bytes(T, Bl, A)
            alone, alone_stats = heads_loss_alone(dev, T, Bl, A)
            if hl_in is None:  # ranks sharing a GPU / elastic / --no-overlap: no one-update pipeline was run beside it
                hl_in, hl_in_stats = alone, None
            b2b = heads_loss_back_to_back(dev, T, Bl, A)
            # the in-pipeline figure is the MEDIAN of the bracketed launches (>= 100 on the full line) with its spread;
            # where the kernel was not run beside the actors (hl_in_stats is None) it is the stand-alone figure
            t_in = (hl_in_stats['median'] * 1e-6) if hl_in_stats else hl_in
            alone_med = alone_stats['median'] * 1e-6
            fr = lambda t: by / t / 1e9 / HBM_PEAK_GBPS   # noqa: E731
            out['roofline'] = {
                'kernel': 'impala_heads_loss_q_kernel (policy_fc + value_fc + log-softmax / entropy / KL + V-trace + loss '
                          'sums + gradient w.r.t. the trunk output and the heads, four waves per sequence, T=%d B=%d A=%d; '
                          'HIP events around the C-ABI call on the learner stream = this kernel + its 4 us '
                          'heads_partial_sum_kernel)' % (T, Bl, A),
                'bound': 'hbm', 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'bytes_per_launch': by,
                # `frac` is the IN-PIPELINE figure: the kernel beside the actors' emulator / MFMA kernels, wherever in
                # their step the two free-running streams happen to put it (no launch-phase tuning): median of n launches
                'achieved': by / t_in / 1e9, 'frac': fr(t_in),
                'frac_in_pipeline': fr(t_in) if hl_in_stats else None,
                'frac_in_pipeline_p10_p90': [fr(hl_in_stats['p90'] * 1e-6), fr(hl_in_stats['p10'] * 1e-6)] if hl_in_stats else None,
                'frac_in_pipeline_of_the_mean': fr(hl_in) if hl_in_stats else None,
                'in_pipeline_us': hl_in_stats,
                'in_pipeline_over_alone': (hl_in_stats['median'] / alone_stats['median']) if hl_in_stats else None,
                'in_pipeline_measured_in': ('impala_one_update leg (this process)' if one is not None else
                                            'the timed region' if hl_in_stats else
                                            'not measured in this configuration: frac is the stand-alone figure'),
                'frac_alone': fr(alone_med), 'achieved_alone': by / alone_med / 1e9, 'alone_us': alone_stats,
                # per call with nothing between the calls (50 C-ABI calls between one event pair, measured in this run):
                # the two kernels' device time without the launch gaps of a single bracketed call
                'back_to_back_us': (b2b * 1e6) if b2b else None, 'frac_back_to_back': fr(b2b) if b2b else None,
                # the main kernel by itself (rocprofv3 --kernel-trace --stats of the committed profile; bench.py cannot
                # run under rocprof itself)
                'kernel_only': kernel_only_profile(by) if (T, Bl, A) == (50, 1024, 6) else None,
                'note': 'the V-trace scan at the WORKLOAD shape (T=50 x 1024 sequences), fused with the two heads so that '
                        'the 52 MB trunk output and its gradient cross HBM once each.  In the headline\'s learner mode '
                        '(train_batch_size 1000 = 20 sequences per update) the same kernel moves %.1f MB per launch '
                        'inside a hipGraph: launch-latency-bound by construction, the 60 %% target does not apply '
                        'there; see roofline_saturating for the bare scan at a saturating shape' %
                        (heads_loss_bytes(T, max(1, args.train_batch // T), A) / 1e6),
            }
            out['roofline'].update(pmc_traffic('impala_heads_loss_T%d_B%d_A%d' % (T, Bl, A)))
        else:
            vt = vt_timer.mean_seconds()
            # 73 B/elt at A=6 for the fused V-trace from logits (2 logits rows, action, reward,
            # done, value in; vs, pg_adv out); the one-kernel loss additionally writes the gradient
            # w.r.t. logits and values (4A + 4 B/elt)
            by = T * Eg * (2 * A * 4 + 8 + 4 + 1 + 4) + (T - 1) * Eg * 8 + T * Eg * (4 * A + 4)
            out['roofline'] = {
                'kernel': 'impala_loss_wave_kernel (V-trace + log-prob gather + entropy + KL + loss sums + gradient, '
                          'wave per sequence, T=%d B=%d A=%d; %d launch(es) per update, one per actor group)' % (T, Eg, A, G),
                'bound': 'hbm', 'achieved': by / vt / 1e9, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                'frac': by / vt / 1e9 / HBM_PEAK_GBPS, 'bytes_per_launch': by,
                'note': 'workload shape is %.1f MB: launch-latency-bound by construction (SURVEY 8d); '
                        'see roofline_saturating for the HBM-bound shape' % (by / 1e6),
            }
            out['roofline'].update(pmc_traffic('impala_loss_T%d_B%d_A%d' % (T, Eg, A)))
        # --- the same scan family at the saturating shape (T'=127, B=262,144: 932 MB) ---
        Ts, Bs = (127, 262144) if not args.quick else (127, 8192)
        x = [torch.randn((Ts, Bs), device=dev) for _ in range(5)]
        boot = torch.randn(Bs, device=dev)
        sat = KernelTimer()
        f = sat.wrap(ops.vtrace)
        for _ in range(3):
            ops.vtrace(x[0], x[1], x[2], x[3], x[4], boot)
        sat.enabled = True
        for _ in range(20):
            f(x[0], x[1], x[2], x[3], x[4], boot)
        torch.cuda.synchronize()
        bys = Ts * Bs * 28 + 4 * Bs
        out['roofline_saturating'] = {
            'kernel': 'vtrace_tm_kernel (from log-probs, lane per sequence, T=%d B=%d%s)' %
                      (Ts, Bs, ': the saturating shape of SURVEY 8d' if not args.quick else ' (--quick: reduced shape)'),
            'bound': 'hbm',
            'achieved': bys / sat.mean_seconds() / 1e9, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
            'frac': bys / sat.mean_seconds() / 1e9 / HBM_PEAK_GBPS, 'bytes_per_launch': bys,
        }
        out['roofline_saturating'].update(pmc_traffic('vtrace_T127_B262144'))
        del x
        fpb = Eg * (2 * 33600 + dim * dim)  # two colour frames read, dim^2 written per env-step
        out['roofline_frame_post'] = {
            'kernel': 'frame_post_kernel + since_update_kernel (max-2, gray, INTER_AREA %dx%d, E=%d)%s' % (
                dim, dim, Eg, '; stand-alone: in the rollout this work is the tail of atari_env_kernel '
                '(parlhip_atari_vec_step_obs), not a launch' if fused_obs else ''),
            'bound': 'hbm', 'achieved': fpb / fps / 1e9, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
            'frac': fpb / fps / 1e9 / HBM_PEAK_GBPS, 'bytes_per_launch': fpb,
        }
        out['roofline_frame_post'].update(pmc_traffic('frame_post_E%d_d%d' % (Eg, dim)))
        # the dominant kernel, for completeness: algorithmic HBM bytes of one VectorEnv.step launch
        # (per env: 512 B state read + written, two 33,600 B colour frames written, action / reward /
        # done / flags) against its duration.  It is NOT HBM-bound: one wavefront per env executes the 6507 of the
        # cartridge serially; what bounds it is single-wave instruction issue.
        # the dominant kernel of the rollout is NOT bandwidth-bound (70 MB of frame stores per launch in ~0.65 ms):
        # one wavefront pair per env executes the cartridge's 6507 serially.  What describes it is how many of the
        # issue slots it occupies it uses — from the committed SQ counter run of this kernel (tools/pmc_env.sh).
        out['roofline_env_kernel'] = {
            'kernel': 'atari_env_kernel<GAME> (VectorEnv.step: 4 emulated frames for each of %d envs)' % Eg,
            'bound': 'instruction issue latency (one wave pair per env: 6507 on the scalar unit | picture)',
            'env_step_ms_event_timed': es * 1e3,
            'waves_per_simd': 2.0 * Eg / 1024,
            # SQ counters of a committed rocprofv3 --pmc run, quoted only for the configuration they were taken at
            **({'issue_slot_utilisation': ENV_PMC['issue_slot_utilisation'],
                'instructions_per_wave_pair_and_frame': ENV_PMC['instructions_per_frame'],
                'source': ENV_PMC['source'], 'measured_in_this_run': False}
               if (args.game, Eg, dim) == (ENV_PMC['game'], ENV_PMC['envs'], ENV_PMC['dim']) else {}),
            'observation_in_the_same_launch': fused_obs,
            'note': 'no HBM roofline fraction is quoted for this kernel; the event-timed call also contains frame_post',
        }
        out['kernels'] = {
            'env_step_ms (atari_env_kernel + frame_post + since_update, one agent step of one %d-env group; %d '
            'groups run concurrently)' % (Eg, G): es * 1e3,
            'note': 'atari_env_kernel is instruction/latency-bound (serial 6507 per wavefront): no roofline fraction',
        }
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(args.game, dim)
            from oracle import py_baselines  # measurement infrastructure (numpy ports pinned on reference fixtures)
            out['cpu_baseline']['scan_kernels'] = py_baselines.time_scan_baselines()
            try:
                out['cpu_baseline']['torch_cpu'] = torch_cpu_baselines()
            except Exception as e:  # measurement infrastructure must never take the line down
                out['cpu_baseline']['torch_cpu'] = {'error': '%s: %s' % (type(e).__name__, e)}
        if world == 1 and not args.quick:
            torch.cuda.empty_cache()
            out.update(extra_legs(dev))
        print(json.dumps(out))


if __name__ == '__main__':
    main()
