"""Golden fixtures for the PPO rows of SURVEY.md §8 (a10, f4), generated FROM THE REFERENCE ITSELF.

Build-container only (reads /root/reference).  Nothing is copied: the reference's classes are
exec'd / imported in place and only numeric inputs and outputs are stored.

    python tests/golden/make_ppo_golden.py

  * vecnormalize.npz — parl/env/mujoco_wrappers.py:73-206 (RunningMeanStd, VecNormalizeEnv,
    update_mean_var_count_from_moments) driven exactly as examples/PPO/env_utils.py:67-115 drives
    one env (step; on done: reset) on a synthetic float64 observation / reward / done stream.
    gym is absent: a 20-line gym.Wrapper stand-in (attribute delegation only) hosts the class.
  * ppo_learn.npz — parl/algorithms/torch/ppo.py:27-206 (PPO.learn, continuous and discrete)
    imported with the 5-stub shim of SURVEY.md A4 on CPU; model = the MLP of
    examples/PPO/mujoco_model.py restated in torch (the reference model file is Paddle); stores
    the initial state_dict, the minibatches, the three returned losses per call and the
    parameters after the updates.
  * ppo_sample_batch.npz — examples/PPO/storage.py:18-76 (append ring, compute_returns,
    sample_batch) on random rollouts with shuffled minibatch indices (agent.py:91-99).
"""
import ast
import collections
import importlib.util
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True  # modules are imported from /root/reference by path: never leave a __pycache__ there

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))


def _exec_defs(path, names, extra_globals):
    src = open(path).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    assert len(keep) == len(names), (path, names)
    g = dict(extra_globals)
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, 'exec'), g)
    return g


def make_vecnormalize():
    gym = types.ModuleType('gym')

    class Wrapper(object):
        def __init__(self, env):
            self.env = env
            self.observation_space = env.observation_space
            self.action_space = getattr(env, 'action_space', None)

        def __getattr__(self, name):
            if name.startswith('__'):
                raise AttributeError(name)
            return getattr(self.env, name)

    gym.Wrapper = Wrapper
    g = _exec_defs(os.path.join(REF, 'parl/env/mujoco_wrappers.py'),
                   ['RunningMeanStd', 'VecNormalizeEnv', 'update_mean_var_count_from_moments'],
                   {'np': np, 'gym': gym})
    VecNormalizeEnv = g['VecNormalizeEnv']
    Space = collections.namedtuple('Space', ['shape'])

    class StreamEnv(object):
        """synthetic host env: pre-drawn float64 observations / rewards / dones"""

        def __init__(self, obs, rew, done, reset_obs):
            self.obs, self.rew, self.done, self.reset_obs = obs, rew, done, reset_obs
            self.t, self.k = 0, 0
            self.observation_space = Space((obs.shape[1], ))

        def step(self, action):
            o, r, d = self.obs[self.t], float(self.rew[self.t]), bool(self.done[self.t])
            self.t += 1
            return o.copy(), r, d, {}

        def reset(self):
            o = self.reset_obs[self.k]
            self.k += 1
            return o.copy()

    rng = np.random.default_rng(2024)
    out = {}
    for name, E, D, S, pdone in [('E6_D17_S60', 6, 17, 60, 0.08), ('E3_D5_S200', 3, 5, 200, 0.02)]:
        raw = rng.standard_normal((S, E, D)) * rng.uniform(0.1, 30.0, (1, 1, D)) + rng.uniform(-5, 5, (1, 1, D))
        rew = rng.standard_normal((S, E)) * 3.0
        done = rng.random((S, E)) < pdone
        rst = rng.standard_normal((S + 1, E, D)) * 0.1
        envs = [VecNormalizeEnv(StreamEnv(raw[:, e], rew[:, e], done[:, e], rst[:, e]), gamma=0.99) for e in range(E)]
        first = np.stack([env.reset() for env in envs])          # ParallelEnv.reset, env_utils.py:62-66
        o_out = np.zeros((S, E, D))
        r_out = np.zeros((S, E))
        o_term = np.zeros((S, E, D))                              # filtered terminal obs (discarded by the caller)
        for t in range(S):
            for e, env in enumerate(envs):                         # ParallelEnv.step, env_utils.py:67-115
                o, r, d, _ = env.step(None)
                o_term[t, e] = o
                if d:
                    o = env.reset()
                o_out[t, e], r_out[t, e] = o, np.asarray(r).reshape(-1)[0]
        out[name + '/raw_obs'], out[name + '/raw_rew'], out[name + '/done'] = raw, rew, done
        out[name + '/reset_obs'] = rst
        out[name + '/first_obs'] = first
        out[name + '/obs'], out[name + '/rew'], out[name + '/obs_terminal'] = o_out, r_out, o_term
        out[name + '/ob_mean'] = np.stack([env.ob_rms.mean for env in envs])
        out[name + '/ob_var'] = np.stack([env.ob_rms.var for env in envs])
        out[name + '/ob_count'] = np.array([env.ob_rms.count for env in envs])
        out[name + '/ret_mean'] = np.array([env.ret_rms.mean for env in envs])
        out[name + '/ret_var'] = np.array([env.ret_rms.var for env in envs])
        out[name + '/ret_count'] = np.array([env.ret_rms.count for env in envs])
        out[name + '/ret'] = np.array([env.ret[0] for env in envs])
    np.savez_compressed(os.path.join(OUT, 'vecnormalize.npz'), **out)
    print('vecnormalize obs[0,0,:3] =', out['E6_D17_S60/obs'][0, 0, :3], 'rew[5,:3] =', out['E6_D17_S60/rew'][5, :3])


def _import_reference_parl():
    """five stubs, torch backend"""
    stubs = {
        'termcolor': "def colored(s, *a, **k):\n    return s\n",
        'pynvml': '',
    }
    for name, src in stubs.items():
        m = types.ModuleType(name)
        exec(src, m.__dict__)
        sys.modules[name] = m
    zmq = types.ModuleType('zmq')
    zmq.Context = type('Context', (), {})
    zmq.REQ = zmq.REP = zmq.RCVTIMEO = zmq.DEALER = zmq.ROUTER = zmq.LINGER = zmq.SNDTIMEO = 0
    zmq.error = types.SimpleNamespace(Again=type('Again', (Exception, ), {}))
    sys.modules['zmq'] = zmq
    fc = types.ModuleType('flask_cors')
    fc.CORS = lambda *a, **k: None
    sys.modules['flask_cors'] = fc

    class _NoPyarrow(object):  # `import pyarrow` must raise ImportError (communication.py:22-26)
        def find_spec(self, name, path=None, target=None):
            if name == 'pyarrow' or name.startswith('pyarrow.'):
                raise ImportError('stubbed out')
            return None

    sys.meta_path.insert(0, _NoPyarrow())
    os.environ['PARL_BACKEND'] = 'torch'
    os.environ['PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION'] = 'python'
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir('/tmp')  # the reference logger may create train_log/
    try:
        import parl  # noqa: F401
        from parl.algorithms import PPO
    finally:
        os.chdir(cwd)
    return parl, PPO


def make_ppo_learn():
    import torch
    import torch.nn as nn
    parl, PPO = _import_reference_parl()
    torch.manual_seed(0)

    class MujocoModel(parl.Model):  # examples/PPO/mujoco_model.py:21-63 in torch
        def __init__(self, obs_dim, act_dim):
            super().__init__()
            self.fc1 = nn.Linear(obs_dim, 64)
            self.fc2 = nn.Linear(64, 64)
            self.fc_value = nn.Linear(64, 1)
            self.fc_policy = nn.Linear(64, act_dim)
            self.fc_pi_std = nn.Parameter(torch.zeros(1, act_dim))

        def value(self, obs):
            return self.fc_value(torch.tanh(self.fc2(torch.tanh(self.fc1(obs)))))

        def policy(self, obs):
            out = torch.tanh(self.fc2(torch.tanh(self.fc1(obs))))
            return self.fc_policy(out), torch.exp(self.fc_pi_std)

    class DiscreteModel(parl.Model):
        def __init__(self, obs_dim, act_dim):
            super().__init__()
            self.fc1 = nn.Linear(obs_dim, 64)
            self.fc_value = nn.Linear(64, 1)
            self.fc_policy = nn.Linear(64, act_dim)

        def value(self, obs):
            return self.fc_value(torch.tanh(self.fc1(obs)))

        def policy(self, obs):
            return self.fc_policy(torch.tanh(self.fc1(obs)))

    out = {}
    rng = np.random.default_rng(7)
    # the reference PPO moves the model to cuda when available; this container has none -> CPU
    for name, cont, obs_dim, act_dim, nb, kw in [
        ('continuous', True, 17, 6, 256, dict(clip_param=0.2, entropy_coef=0.0, initial_lr=3e-4)),
        ('discrete', False, 11, 4, 128, dict(clip_param=0.1, entropy_coef=0.01, initial_lr=2.5e-4)),
        ('continuous_noclipv_nonorm', True, 5, 2, 64,
         dict(clip_param=0.2, entropy_coef=0.0, initial_lr=1e-3, use_clipped_value_loss=False, norm_adv=False)),
    ]:
        model = (MujocoModel if cont else DiscreteModel)(obs_dim, act_dim)
        alg = PPO(model, continuous_action=cont, **kw)
        for k, v in model.state_dict().items():
            out['%s/init/%s' % (name, k)] = v.detach().numpy().copy()
        losses = []
        for it in range(3):
            obs = rng.standard_normal((nb, obs_dim)).astype(np.float32)
            if cont:
                act = rng.standard_normal((nb, act_dim)).astype(np.float32)
            else:
                act = rng.integers(0, act_dim, nb).astype(np.int64)
            val = rng.standard_normal(nb).astype(np.float32)
            ret = (val + rng.standard_normal(nb) * 0.5).astype(np.float32)
            logp = (rng.standard_normal(nb) * 0.3 - (6.0 if cont else 1.3)).astype(np.float32)
            adv = (rng.standard_normal(nb) * 2.0 + 0.3).astype(np.float32)
            lr = None if it == 0 else 2e-4 / it
            res = alg.learn(torch.from_numpy(obs), torch.from_numpy(act), torch.from_numpy(val), torch.from_numpy(ret),
                            torch.from_numpy(logp), torch.from_numpy(adv), lr)
            losses.append(res)
            for k, v in dict(obs=obs, act=act, val=val, ret=ret, logp=logp, adv=adv).items():
                out['%s/batch%d/%s' % (name, it, k)] = v
            out['%s/batch%d/lr' % (name, it)] = np.array(np.nan if lr is None else lr)
        out[name + '/losses'] = np.array(losses, np.float64)
        for k, v in model.state_dict().items():
            out['%s/final/%s' % (name, k)] = v.detach().numpy().copy()
        out[name + '/dims'] = np.array([obs_dim, act_dim, nb])
        out[name + '/kw'] = np.array([kw['clip_param'], kw['entropy_coef'], kw['initial_lr'],
                                      float(kw.get('use_clipped_value_loss', True)), float(kw.get('norm_adv', True))])
    np.savez_compressed(os.path.join(OUT, 'ppo_learn.npz'), **out)
    print('ppo_learn continuous losses =', out['continuous/losses'])


def make_sample_batch():
    spec = importlib.util.spec_from_file_location('ref_ppo_storage', os.path.join(REF, 'examples/PPO/storage.py'))
    st = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(st)
    Space = collections.namedtuple('Space', ['shape'])
    rng = np.random.default_rng(5)
    out = {}
    for name, T, E, od, ad in [('T12_E6', 12, 6, (17, ), (6, )), ('T9_E4_discrete', 9, 4, (5, ), ())]:
        rs = st.RolloutStorage(T, E, Space(od), Space(ad))
        steps = T + 5  # the ring wraps: cur_step = (cur_step + 1) % step_nums (storage.py:43)
        rows = []
        for t in range(steps):
            row = (rng.standard_normal((E, ) + od).astype(np.float32),
                   (rng.standard_normal((E, ) + ad) if ad else rng.integers(0, 4, E)).astype(np.float32),
                   rng.standard_normal(E).astype(np.float32), rng.standard_normal(E).astype(np.float32),
                   (rng.random(E) < 0.2).astype(np.float32), rng.standard_normal(E).astype(np.float32))
            rs.append(*row)
            rows.append(row)
        value = rng.standard_normal(E).astype(np.float32)
        done = (rng.random(E) < 0.3).astype(np.float32)
        rs.compute_returns(value, done)
        idx = np.arange(T * E)
        rng.shuffle(idx)
        idx = idx[:T * E // 3]
        b = rs.sample_batch(idx)
        for i, k in enumerate(['obs', 'actions', 'logprobs', 'rewards', 'dones', 'values']):
            out['%s/append_%s' % (name, k)] = np.stack([r[i] for r in rows])
        out[name + '/value'], out[name + '/done'], out[name + '/idx'] = value, done, idx
        for i, k in enumerate(['obs', 'actions', 'logprobs', 'advantages', 'returns', 'values']):
            out['%s/batch_%s' % (name, k)] = b[i]
        out[name + '/cur_step'] = np.array(rs.cur_step)
    np.savez_compressed(os.path.join(OUT, 'ppo_sample_batch.npz'), **out)
    print('sample_batch adv[:3] =', out['T12_E6/batch_advantages'][:3])


if __name__ == '__main__':
    if not os.path.isdir(REF):
        sys.exit('needs /root/reference (build container only)')
    make_vecnormalize()
    make_sample_batch()
    make_ppo_learn()
