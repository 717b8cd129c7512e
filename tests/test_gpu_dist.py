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
