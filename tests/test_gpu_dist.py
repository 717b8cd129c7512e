"""RCCL for real on the one GPU of the test box, and bench.py's self-launch.  -m gpu."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_rccl_one_rank_group_runs_every_collective_of_the_dp_path():
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('WORLD_SIZE', None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'tools', 'rccl_ws1_check.py'), str(_free_port())],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, text=True)
    assert p.returncode == 0 and 'RCCL_WS1_OK' in p.stdout, p.stdout[-4000:]


def test_ranks_sharing_one_gpu_all_reduce_through_ipc_slots():
    """parl_amd.dist.SharedDeviceAllReduce: two ranks on this box's one GPU exchange the gradient bucket through HIP
    IPC slots (what lets a two-rank data-parallel run LEARN on a one-GPU box at hundreds of updates/s instead of
    gloo's host-staged ~25): every rank ends with the sum in rank order, bit-identical, six calls in a row"""
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY='0', WORLD_SIZE='2', RANK=str(r),
                   LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), PARL_AMD_SHARE_GPU='1',
                   PARL_AMD_DIST_BACKEND='gloo')
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'tools', 'shared_allreduce_check.py')],
                                      cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs) and all('SHARED_ALLREDUCE_OK' in o for o in outs), [o[-2000:] for o in outs]


def test_bench_self_launches_two_ranks_without_torchrun():
    """`python bench.py --gpus 2` (the form the driver uses for N=1) must start its own ranks; on a
    1-GPU box they share the device over gloo and the JSON line says so."""
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'PARL_AMD_SHARE_GPU', 'PARL_AMD_DIST_BACKEND'):
        env.pop(k, None)
    cmd = [sys.executable, 'bench.py', '--gpus', '2', '--steps', '2', '--warmup', '1', '--envs', '64',
           '--sample-batch-steps', '10', '--no-cpu-baseline', '--quick']
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, text=True)
    assert p.returncode == 0, p.stdout[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(line) == 1, p.stdout[-2000:]
    out = json.loads(line[0])
    assert out['n_gpus'] == 2 and out['value'] > 0 and out['config']['train_batch'] == 2 * 64 * 10
    import torch
    if torch.cuda.device_count() < 2:
        assert 'SHARE' in out['config']['collectives']


def test_bench_data_parallel_path_over_rccl_with_one_rank():
    """PARL_AMD_FORCE_DIST=1: bench.py creates a one-rank RCCL group and runs its complete
    data-parallel path (weight broadcast, flat-gradient all-reduce + trajectory all-gather on the
    learner stream inside the overlapped pipeline, barrier, max-over-ranks timing) on the real backend."""
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY='0', PARL_AMD_FORCE_DIST='1',
               MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'PARL_AMD_SHARE_GPU', 'PARL_AMD_DIST_BACKEND'):
        env.pop(k, None)
    cmd = [sys.executable, 'bench.py', '--gpus', '1', '--steps', '2', '--warmup', '1', '--envs', '64',
           '--sample-batch-steps', '10', '--no-cpu-baseline', '--quick']
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, text=True)
    assert p.returncode == 0, p.stdout[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')][0])
    assert out['n_gpus'] == 1 and out['value'] > 0 and 'RCCL' in out['config']['collectives']


def test_bench_reference_train_batch_over_rccl_with_one_rank():
    """the hipGraph updates in their data-parallel form (forward + backward graph | RCCL all-reduce | clip + Adam
    graph) on the REAL backend: PARL_AMD_FORCE_DIST=1 creates a one-rank RCCL group, --train-batch makes every
    rollout several graph-replayed updates with the all-reduce between the two graphs on the learner stream."""
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY='0', PARL_AMD_FORCE_DIST='1',
               MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'PARL_AMD_SHARE_GPU', 'PARL_AMD_DIST_BACKEND'):
        env.pop(k, None)
    cmd = [sys.executable, 'bench.py', '--gpus', '1', '--steps', '3', '--warmup', '1', '--envs', '32',
           '--sample-batch-steps', '10', '--train-batch', '80', '--no-cpu-baseline', '--quick']
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, text=True)
    assert p.returncode == 0, p.stdout[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    out = json.loads(line[0])
    assert 'RCCL' in out['config']['collectives'] and out['config']['learner_updates_per_step'] == 4
    assert out['learner_updates_per_sec'] > 0 and out['value'] > 0
    # round 6: the collective is captured INSIDE the update's graph (one launch per update), and the line says so; the
    # all-reduce of the real bucket is event-timed after the timed region; per-rank figures ride along
    assert out['config']['dp_update_form'].startswith('ONE hipGraph per update'), out['config']['dp_update_form']
    assert out['grad_allreduce_alone']['n'] == 30 and out['grad_allreduce_alone']['median_us'] > 0
    assert len(out['per_rank']['env_frames_per_s']) == 1


def test_update_graph_falls_back_to_two_graphs_when_the_collective_is_not_captured():
    """PARL_AMD_GRAPH_ALLREDUCE=0 (and any backend whose collective cannot be captured): forward + backward graph | eager
    all-reduce | clip + Adam graph — round 3's form, still selectable, same results path"""
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY='0', PARL_AMD_FORCE_DIST='1', PARL_AMD_GRAPH_ALLREDUCE='0',
               MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'PARL_AMD_SHARE_GPU', 'PARL_AMD_DIST_BACKEND'):
        env.pop(k, None)
    cmd = [sys.executable, 'bench.py', '--gpus', '1', '--steps', '3', '--warmup', '1', '--envs', '32',
           '--sample-batch-steps', '10', '--train-batch', '80', '--no-cpu-baseline', '--quick']
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, text=True)
    assert p.returncode == 0, p.stdout[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')][0])
    assert out['config']['dp_update_form'].startswith('two hipGraphs per update'), out['config']['dp_update_form']
    assert out['learner_updates_per_sec'] > 0 and out['value'] > 0


def _bench_env(**kw):
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(_free_port()))
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'PARL_AMD_SHARE_GPU', 'PARL_AMD_DIST_BACKEND', 'PARL_AMD_FORCE_DIST'):
        env.pop(k, None)
    env.update({k: str(v) for k, v in kw.items()})
    return env


def _error_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith('{"error"')]
    assert len(lines) == 1, stdout[-3000:]
    return json.loads(lines[0])


def test_bench_rank_whose_peer_never_shows_up_reports_and_exits():
    """The first real N > 1 run must not be able to hang silently: rank 0 of a two-rank RCCL job whose rank 1 was
    never started gives up after the process-group timeout, prints ONE JSON error line (rank, what it waited for)
    and exits non-zero."""
    import time
    t0 = time.time()
    env = _bench_env(WORLD_SIZE=2, RANK=0, LOCAL_RANK=0, PARL_AMD_DIST_TIMEOUT=8)
    cmd = [sys.executable, 'bench.py', '--gpus', '2', '--steps', '2', '--warmup', '1', '--envs', '64',
           '--sample-batch-steps', '10', '--no-cpu-baseline', '--quick']
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300, text=True)
    assert p.returncode == 2, (p.returncode, p.stdout[-3000:])
    err = _error_line(p.stdout)
    assert err['rank'] == 0 and err['world_size_env'] == 2 and 'rendezvous failed' in err['error'], err
    assert time.time() - t0 < 120


def test_bench_watchdog_ends_a_rank_that_makes_no_progress():
    """--hang-timeout: no beat for that long (here: a limit shorter than the set-up itself) -> one JSON error line
    naming the phase, exit code 3 — what a collective that a dead peer never joins looks like from the outside."""
    cmd = [sys.executable, 'bench.py', '--gpus', '1', '--steps', '2', '--warmup', '1', '--envs', '64',
           '--sample-batch-steps', '10', '--no-cpu-baseline', '--quick', '--hang-timeout', '0.5']
    p = subprocess.run(cmd, cwd=ROOT, env=_bench_env(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300,
                       text=True)
    assert p.returncode == 3, (p.returncode, p.stdout[-3000:])
    err = _error_line(p.stdout)
    assert 'no progress' in err['error'] and err['rank'] == 0


def test_two_rank_data_parallel_impala_training_runs_in_step(capsys):
    """examples/IMPALA/train.py as a two-rank data-parallel job (the ranks share this box's GPU, so gloo instead of
    RCCL — every update then costs a host-staged 4 MB all-reduce, ~25 updates/s, far too few to see Pong's score
    move; the LEARNING check of the data-parallel semantics is tests/test_dist_gloo.py::
    test_two_rank_data_parallel_training_learns_cartpole).  What this run pins: both ranks take exactly the same
    number of updates (the stop decision is a max over ranks), log finite losses, sample different envs
    (env ids rank * E ..), and end cleanly."""
    import ast
    import re
    port = _free_port()
    procs = []
    for r in range(2):
        env = _bench_env(WORLD_SIZE=2, RANK=r, LOCAL_RANK=r, MASTER_PORT=port, PARL_AMD_SHARE_GPU=1,
                         PARL_AMD_DIST_BACKEND='gloo')
        procs.append(subprocess.Popen([sys.executable, 'examples/IMPALA/train.py', '--minutes', '0.3', '--env-num', '16',
                                       '--log-interval', '6', '--seed', '1'], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[-2000:] for o in outs]
    curves = []
    for o in outs:
        rows = [ast.literal_eval(m.group(1)) for m in re.finditer(r"INFO\] (\{'sample_steps'.*\})\s*$", o, re.M)]
        assert rows, o[-2000:]
        curves.append([(r['elapsed_time_s'], r['mean_episode_rewards'], r['learn_steps'], r.get('total_loss')) for r in rows])
    with capsys.disabled():
        print('\nIMPALA Pong, 2 ranks x 16 envs, DP on one GPU (elapsed s, mean_episode_rewards, updates, loss):', curves)
    assert curves[0][-1][2] == curves[1][-1][2] and curves[0][-1][2] >= 100   # the same number of updates on both ranks
    import math
    for c in curves:
        assert all(x[3] is None or math.isfinite(x[3]) for x in c), c
    assert [x[1] for x in curves[0]] != [x[1] for x in curves[1]]   # different envs on the two ranks
