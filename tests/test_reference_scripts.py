"""(1) The reference's OWN headline examples — examples/IMPALA/{train, actor, atari_model, atari_agent, impala_config}.py
and examples/A2C/{...}.py, Paddle-flavoured — executed UNMODIFIED through compat/{paddle,parl,gym} in a subprocess
(tests/tools/run_reference_example.py: their actor / learner threads never stop): `import paddle` resolves to the
torch-backed alias of exactly the Paddle surface those files touch, `parl` / `gym` to parl_amd / the device env.
Learner with its learn thread and DataLoader.from_generator reader, @parl.remote_class Actors stepping
VectorEnv([wrap_deepmind(gym.make(...))]), np.random.choice sampling, parl.algorithms.IMPALA / A2C learn,
get_weights / set_weights parameter sync, schedulers, MonitorEnv metrics: all the example's own code.

(2) The reference's OWN torch A2C example — benchmark/torch/a2c/{train.py, actor.py, atari_agent.py,
atari_model.py, a2c_config.py} — executed UNMODIFIED through the drop-in boundary:
`import gym` / `import parl` resolve to compat/gym and compat/parl (aliases of parl_amd), and
    gym.make -> wrap_deepmind(dim, obs_format='NCHW') -> VectorEnv(envs).reset()/step()
    @parl.remote_class(wait=False) Actor, parl.connect, future .get()
    parl.Model / parl.Agent / parl.algorithms.A2C(model, config) / get_weights / set_weights
    parl.utils.rl_utils.calc_gae per (env, segment), MonitorEnv.next_episode_results via
    get_wrapper_by_cls, LinearDecayScheduler / PiecewiseScheduler, WindowStat / TimeStat,
    logger / summary
are exactly the names, signatures and return types the script uses (actor.py:30-122,
train.py:33-178).  Only the config dict is shrunk (data, not code).

Two variants of the same body:
  * CPU (runs in the build container, where /root/reference exists): the two GPU-only pieces are
    replaced by TEST DOUBLES backed by the CPU oracle — DeviceVectorEnv by the oracle's VecEnv and
    calc_gae by the oracle's scan — so the script, the host layer (handles, VectorEnv, MonitorEnv
    bookkeeping, remote proxies, Agent / Algorithm / Model, A2C.learn) run for real;
  * -m gpu: the real device path (env kernel, frame_post, GAE kernel).  The GPU boxes have no
    /root/reference, and reference sources are never committed: build() (oracle/make_ref.py) stages
    the five scripts byte for byte into the git-ignored oracle/_ref/a2c/, which travels with the
    snapshot like the built .so files, and this test runs them from there.
"""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

REF_DIR = '/root/reference/benchmark/torch/a2c'
STAGED_DIR = os.path.join(ROOT, 'oracle', '_ref', 'a2c')  # oracle/make_ref.py, git-ignored
SCRIPTS = ['train.py', 'actor.py', 'atari_agent.py', 'atari_model.py', 'a2c_config.py']


def _script_dir():
    for d in (REF_DIR, STAGED_DIR):
        if all(os.path.exists(os.path.join(d, s)) for s in SCRIPTS):
            return d
    pytest.skip('needs the reference a2c scripts (/root/reference or oracle/_ref/a2c staged by build())')


def test_staged_scripts_are_the_reference_files_byte_for_byte():
    if not os.path.isdir(REF_DIR):
        pytest.skip('needs /root/reference (build container)')
    if not os.path.isdir(STAGED_DIR):
        pytest.skip('oracle/_ref/a2c not staged (python __graft_entry__.py)')
    for s in SCRIPTS:
        assert open(os.path.join(STAGED_DIR, s), 'rb').read() == open(os.path.join(REF_DIR, s), 'rb').read(), s


def _run_reference_a2c(script_dir, monkeypatch, steps, env_num, actor_num, T):
    monkeypatch.syspath_prepend(os.path.join(ROOT, 'compat'))
    monkeypatch.syspath_prepend(script_dir)
    for m in ['gym', 'parl', 'train', 'actor', 'atari_agent', 'atari_model', 'a2c_config']:
        monkeypatch.delitem(sys.modules, m, raising=False)
    monkeypatch.chdir('/tmp')  # the script's logger may create train_log/ in the cwd
    import parl_amd
    import gym  # compat/gym
    import parl  # compat/parl
    assert parl is parl_amd and gym.make.__module__ == 'gym'
    train = importlib.import_module('train')
    assert os.path.dirname(os.path.abspath(train.__file__)) == os.path.abspath(script_dir)
    cfg = dict(importlib.import_module('a2c_config').config)
    cfg.update(actor_num=actor_num, env_num=env_num, sample_batch_steps=T, max_sample_steps=10**6)
    learner = train.Learner(cfg)
    w0 = {k: v.copy() for k, v in learner.agent.get_weights().items()}
    for _ in range(steps):
        learner.step()
    assert learner.sample_total_steps == steps * actor_num * env_num * T
    assert learner.should_stop() is False
    for stat in (learner.total_loss_stat, learner.pi_loss_stat, learner.vf_loss_stat, learner.entropy_stat):
        assert stat.count == steps and np.isfinite(stat.mean)
    assert learner.lr is not None and 0 < learner.lr <= cfg['start_lr'] and learner.entropy_coeff == -0.01
    w1 = learner.agent.get_weights()
    assert any(np.abs(w1[k] - w0[k]).max() > 0 for k in w0), 'the learner updated nothing'
    # actors hold the learner's weights after set_weights (train.py:82-85)
    learner.step()
    learner.log_metrics()  # MonitorEnv statistics through get_wrapper_by_cls
    ms = [a.get_metrics().get() for a in learner.remote_actors]
    assert all(isinstance(m, dict) for m in ms)
    return learner


class _OracleVecEnvDouble(object):
    """test double for parl_amd.env.device_vector_env.DeviceVectorEnv on a box without a GPU"""

    def __init__(self, env_name, num_envs, dim=84, horizon=1, seed=0, env_id0=0, device=None, **kw):
        from oracle import c_oracle
        from parl_amd.env import GAMES, find_rom
        name = GAMES[env_name][0]
        self.v = c_oracle.VecEnv(find_rom(name), name, num_envs, dim, seed=seed, env_id0=env_id0)
        self.envs_num, self.dim, self.act_dim = num_envs, dim, self.v.num_actions
        self.device = torch.device('cpu')

    def reset(self):
        return torch.from_numpy(self.v.reset())

    def step(self, actions):
        obs, rew, done = self.v.step(actions.numpy())
        ret = np.zeros(self.envs_num, np.float32)
        ln = np.zeros(self.envs_num, np.int32)
        for e in range(self.envs_num):
            eps = self.v.pop_episodes(e)
            if eps:
                ret[e], ln[e] = eps[-1]
        info = {'episode_returns': torch.from_numpy(ret), 'episode_lengths': torch.from_numpy(ln)}
        return torch.from_numpy(obs), torch.from_numpy(rew), torch.from_numpy(done.astype(bool)), info

    def check_faults(self):
        pass


def _oracle_calc_gae(rewards, values, next_value, gamma, lam):
    from oracle import c_oracle
    r = np.asarray(rewards, np.float32).reshape(-1, 1)
    v = np.asarray(values, np.float32).reshape(-1, 1)
    adv, _ = c_oracle.gae(r, v, np.zeros(r.shape, np.uint8), np.asarray(next_value, np.float32).reshape(-1)[:1],
                          gamma, lam)
    return adv.reshape(-1).astype(np.float64)


def test_reference_torch_a2c_scripts_run_unmodified_on_cpu_doubles(tmp_path, monkeypatch):
    if not os.path.isdir(REF_DIR):
        pytest.skip('needs /root/reference (build container)')
    if not os.path.exists(os.path.join(ROOT, 'roms', 'pong.bin')):
        pytest.skip('cartridge not provisioned')
    import parl_amd.env.vector_env as ve
    import parl_amd.utils.rl_utils as ru
    monkeypatch.setattr(ve, 'DeviceVectorEnv', _OracleVecEnvDouble)
    monkeypatch.setattr(ru, 'calc_gae', _oracle_calc_gae)
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: False)
    torch.set_num_threads(4)
    _run_reference_a2c(REF_DIR, monkeypatch, steps=2, env_num=2, actor_num=2, T=5)


@pytest.mark.gpu
def test_reference_torch_a2c_scripts_run_unmodified_on_the_device(dev, tmp_path, monkeypatch):
    d = _script_dir()
    # the script's own ActorCritic uses nn.Conv2d; this image has no MIOpen kernel database for
    # gfx950 (every new shape would JIT for minutes): use torch's native convolution instead
    monkeypatch.setattr(torch.backends.cudnn, 'enabled', False)
    learner = _run_reference_a2c(d, monkeypatch, steps=3, env_num=4, actor_num=2, T=20)
    assert str(learner.device) == 'cuda'


# ---- the reference's own (Paddle-flavoured) headline examples, unmodified, in a subprocess ----
EXAMPLES = {'impala': ('/root/reference/examples/IMPALA', os.path.join(ROOT, 'oracle', '_ref', 'impala'),
                       ['train.py', 'actor.py', 'atari_agent.py', 'atari_model.py', 'impala_config.py']),
            'a2c_paddle': ('/root/reference/examples/A2C', os.path.join(ROOT, 'oracle', '_ref', 'a2c_paddle'),
                           ['train.py', 'actor.py', 'atari_agent.py', 'atari_model.py', 'a2c_config.py'])}


def _example_dir(kind, prefer_reference=True):
    ref, staged, files = EXAMPLES[kind]
    for d in ((ref, staged) if prefer_reference else (staged, ref)):
        if all(os.path.exists(os.path.join(d, f)) for f in files):
            return d
    pytest.skip('needs the reference %s example (/root/reference or oracle/_ref staged by build())' % kind)


def _run_example(kind, script_dir, extra, timeout):
    import json
    import subprocess
    env = dict(os.environ, PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'tools', 'run_reference_example.py'), kind,
                        script_dir] + extra, cwd='/tmp', env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=timeout, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('RESULT ')]
    assert p.returncode == 0 and len(lines) == 1, p.stdout[-4000:]
    return json.loads(lines[0][7:])


@pytest.mark.parametrize('kind', ['impala', 'a2c_paddle'])
def test_staged_examples_are_the_reference_files_byte_for_byte(kind):
    ref, staged, files = EXAMPLES[kind]
    if not os.path.isdir(ref) or not os.path.isdir(staged):
        pytest.skip('needs /root/reference and the staged copy (python __graft_entry__.py)')
    for f in files:
        assert open(os.path.join(staged, f), 'rb').read() == open(os.path.join(ref, f), 'rb').read(), f


@pytest.mark.parametrize('kind', ['impala', 'a2c_paddle'])
def test_reference_paddle_examples_run_unmodified_on_cpu_doubles(kind):
    if not os.path.exists(os.path.join(ROOT, 'roms', 'pong.bin')):
        pytest.skip('cartridge not provisioned')
    out = _run_example(kind, _example_dir(kind), ['--cpu-doubles', '--steps', '2'], 600)
    assert out['learn_steps'] >= 2 and out['weights_changed'] and np.isfinite(out['total_loss'])
    assert out['lr'] > 0 and out['entropy_coeff'] == -0.01 and out['device'] == 'cpu'
    if kind == 'impala':  # 2 actors x 2 envs x 10 steps per actor batch = the train batch of 40 rows
        assert out['sample_total_steps'] >= 2 * 40 and out['total_params_sync'] >= 2 and np.isfinite(out['kl'])


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['impala', 'a2c_paddle'])
def test_reference_paddle_examples_run_unmodified_on_the_device(kind):
    out = _run_example(kind, _example_dir(kind, prefer_reference=False), ['--steps', '4'], 900)
    assert out['learn_steps'] >= 4 and out['weights_changed'] and np.isfinite(out['total_loss'])
    assert out['device'].startswith('cuda')


def test_running_the_reference_leaves_no_bytecode_in_its_tree():
    """The reference tree is read-only for this project.  Everything here that executes reference modules by path
    (this file, tests/tools/run_reference_example.py, tests/golden/make_*.py, oracle/py_baselines.py) switches
    bytecode writing off, in this process and — through PYTHONDONTWRITEBYTECODE — in the subprocesses it
    spawns; pytest runs this module's tests in file order, so the imports above have happened by now."""
    if not os.path.isdir('/root/reference'):
        pytest.skip('needs /root/reference (build container)')
    assert sys.dont_write_bytecode and os.environ.get('PYTHONDONTWRITEBYTECODE') == '1'
    left = [os.path.join(d, n) for d, sub, files in os.walk('/root/reference')
            for n in sub + files if n == '__pycache__' or n.endswith('.pyc')]
    assert not left, left
