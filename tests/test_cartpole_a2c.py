"""BASELINE.json configs[0]: "CartPole-v1 A2C, 4 CPU remote actors via parl.remote, CPU learner (reference
plumbing, no GPU)" — the structure of the reference's A2C example (examples/A2C/train.py:60-116: kick off all
actors, set_weights(get_weights()), collect the futures, concatenate, ONE learn; examples/A2C/actor.py:51-101:
T steps of a VectorEnv, per-(env, segment) calc_gae at a done or at the end of the rollout, next_value = 0 after
a terminal step) on the host mirror: `@parl.remote_class(wait=False)` actors (in-process futures), `parl.Model`
/ `parl.Agent`, `parl.algorithms.A2C` on torch-CPU, `LinearDecayScheduler`.  It learns: the mean return of the
episodes the actors close reaches >= 150 within 200 updates.

`calc_gae` is the one GPU-only piece of this path (parl_amd.utils.rl_utils is the HIP scan, no CPU fallback): the
CPU variant uses the C oracle's GAE as its test double — exactly as tests/test_reference_scripts.py does — and the
-m gpu variant runs the same loop with the product's calc_gae (the gfx950 kernel) under the actors."""
import os
import sys
from collections import defaultdict

import numpy as np
import pytest
import torch

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, 'tests', 'tools'))
import parl_amd as parl  # noqa: E402
from cartpole import HostVectorEnv, MonitoredCartPole  # noqa: E402
from parl_amd.algorithms import A2C  # noqa: E402
from parl_amd.utils.scheduler import LinearDecayScheduler  # noqa: E402

CONFIG = dict(master_address='localhost:8010', env_name='CartPole-v1', actor_num=4, env_num=4, sample_batch_steps=20,
              gamma=0.99, vf_loss_coeff=0.5, start_lr=0.01, max_sample_steps=3 * 200 * 4 * 4 * 20, entropy_coeff=-0.01)
CONFIG['lambda'] = 1.0


class CartPoleModel(parl.Model):
    def __init__(self, obs_dim, act_dim):
        super(CartPoleModel, self).__init__()
        self.fc1 = torch.nn.Linear(obs_dim, 64)
        self.fc2 = torch.nn.Linear(64, 64)
        self.policy_fc = torch.nn.Linear(64, act_dim)
        self.value_fc = torch.nn.Linear(64, 1)

    def _trunk(self, obs):
        return torch.tanh(self.fc2(torch.tanh(self.fc1(obs))))

    def policy(self, obs):
        return self.policy_fc(self._trunk(obs))

    def value(self, obs):
        return self.value_fc(self._trunk(obs)).squeeze(1)

    def policy_and_value(self, obs):
        h = self._trunk(obs)
        return self.policy_fc(h), self.value_fc(h).squeeze(1)


class CartPoleAgent(parl.Agent):
    """examples/A2C/atari_agent.py:20-110 without the image cast"""

    def __init__(self, algorithm, config, seed=0):
        super(CartPoleAgent, self).__init__(algorithm)
        self.lr_scheduler = LinearDecayScheduler(config['start_lr'], config['max_sample_steps'])
        self.entropy_coeff = config['entropy_coeff']
        self.rng = np.random.default_rng(seed)

    def sample(self, obs_np):
        probs, values = self.alg.prob_and_value(torch.from_numpy(obs_np))
        probs = probs.numpy().astype(np.float64)
        probs /= probs.sum(1, keepdims=True)
        return np.array([self.rng.choice(len(p), p=p) for p in probs]), values.numpy()

    def value(self, obs_np):
        return self.alg.value(torch.from_numpy(obs_np)).numpy()

    def learn(self, obs_np, actions_np, advantages_np, target_values_np):
        lr = self.lr_scheduler.step(step_num=obs_np.shape[0])
        out = self.alg.learn(torch.from_numpy(obs_np), torch.from_numpy(actions_np),
                             torch.from_numpy(advantages_np.astype(np.float32)),
                             torch.from_numpy(target_values_np.astype(np.float32)), lr, self.entropy_coeff)
        return [float(x) for x in out] + [lr]


def make_actor_class(calc_gae):
    @parl.remote_class(wait=False)
    class Actor(object):
        def __init__(self, config, seed):
            self.config = config
            self.envs = [MonitoredCartPole(seed=seed * 100 + i) for i in range(config['env_num'])]
            self.vector_env = HostVectorEnv(self.envs)
            self.obs_batch = self.vector_env.reset()
            model = CartPoleModel(MonitoredCartPole.obs_dim, MonitoredCartPole.act_dim)
            self.agent = CartPoleAgent(A2C(model, vf_loss_coeff=config['vf_loss_coeff']), config, seed=seed)

        def sample(self):
            cfg, sample_data = self.config, defaultdict(list)
            per_env = [defaultdict(list) for _ in range(cfg['env_num'])]
            for i in range(cfg['sample_batch_steps']):
                actions, values = self.agent.sample(np.stack(self.obs_batch))
                next_obs, rewards, dones, _ = self.vector_env.step(actions)
                for e in range(cfg['env_num']):
                    d = per_env[e]
                    d['obs'].append(self.obs_batch[e])
                    d['actions'].append(actions[e])
                    d['rewards'].append(rewards[e])
                    d['values'].append(values[e])
                    if dones[e] or i == cfg['sample_batch_steps'] - 1:  # a segment ends: actor.py:73-85
                        next_value = 0.0 if dones[e] else float(self.agent.value(next_obs[e][None])[0])
                        adv = calc_gae(d['rewards'], d['values'], next_value, cfg['gamma'], cfg['lambda'])
                        sample_data['obs'].extend(d['obs'])
                        sample_data['actions'].extend(d['actions'])
                        sample_data['advantages'].extend(adv)
                        sample_data['target_values'].extend(adv + np.asarray(d['values']))
                        per_env[e] = defaultdict(list)
                self.obs_batch = next_obs
            return {k: np.stack(v) for k, v in sample_data.items()}

        def get_metrics(self):
            m = defaultdict(list)
            for env in self.envs:
                for ret, steps in env.next_episode_results():
                    m['episode_rewards'].append(ret)
                    m['episode_steps'].append(steps)
            return m

        def set_weights(self, params):
            self.agent.set_weights(params)

    return Actor


class Learner(object):
    def __init__(self, config, calc_gae):
        self.config = config
        model = CartPoleModel(MonitoredCartPole.obs_dim, MonitoredCartPole.act_dim)
        self.agent = CartPoleAgent(A2C(model, vf_loss_coeff=config['vf_loss_coeff']), config)
        parl.connect(config['master_address'])
        Actor = make_actor_class(calc_gae)
        self.remote_actors = [Actor(config, seed=config.get('actor_seed0', 1) + i) for i in range(config['actor_num'])]
        self.sample_total_steps, self.updates = 0, 0

    def step(self):
        latest = self.agent.get_weights()
        for a in self.remote_actors:
            a.set_weights(latest)
        futures = [a.sample() for a in self.remote_actors]  # all four sample concurrently (wait=False)
        batch = defaultdict(list)
        for f in futures:
            for k, v in f.get().items():
                batch[k].append(v)
        batch = {k: np.concatenate(v) for k, v in batch.items()}
        self.sample_total_steps += len(batch['obs'])
        self.updates += 1
        return self.agent.learn(batch['obs'], batch['actions'], batch['advantages'], batch['target_values'])

    def episode_returns(self):
        out = []
        for f in [a.get_metrics() for a in self.remote_actors]:
            out.extend(f.get()['episode_rewards'])
        return out


def _oracle_calc_gae(rewards, values, next_value, gamma, lam):
    from oracle import c_oracle
    r = np.asarray(rewards, np.float32).reshape(-1, 1)
    v = np.asarray(values, np.float32).reshape(-1, 1)
    adv, _ = c_oracle.gae(r, v, np.zeros(r.shape, np.uint8), np.asarray(next_value, np.float32).reshape(-1)[:1], gamma, lam)
    return adv.reshape(-1).astype(np.float64)


def _train(calc_gae, max_updates=200, target=150.0):
    torch.manual_seed(0)
    n0 = torch.get_num_threads()
    torch.set_num_threads(1)  # 2-layer MLP on 320 rows: threads only add latency
    try:
        learner = Learner(dict(CONFIG), calc_gae)
        rows = CONFIG['actor_num'] * CONFIG['env_num'] * CONFIG['sample_batch_steps']
        recent, curve = [], []
        for u in range(max_updates):
            losses = learner.step()
            assert np.isfinite(losses).all()
            recent = (recent + learner.episode_returns())[-40:]
            if (u + 1) % 20 == 0 and recent:
                curve.append((u + 1, float(np.mean(recent))))
            if len(recent) >= 20 and np.mean(recent) >= target:
                break
        assert learner.sample_total_steps == learner.updates * rows
        sys.__stdout__.write('\nCartPole-v1 A2C, 4 remote actors x 4 envs: (update, mean return of the last <= 40 episodes) %s '
                             '-> %.1f after %d updates\n' % (curve, np.mean(recent), learner.updates))
        return float(np.mean(recent)), learner.updates
    finally:
        torch.set_num_threads(n0)


def test_cartpole_a2c_four_remote_actors_cpu_learner():
    ret, updates = _train(_oracle_calc_gae)
    assert ret >= 150.0 and updates <= 200, (ret, updates)


@pytest.mark.gpu
def test_cartpole_a2c_four_remote_actors_product_calc_gae(dev):
    """the same loop with the PRODUCT's calc_gae under the actors: parl_amd.utils.calc_gae = the gfx950 GAE
    kernel, one (env, segment) per call as the reference's actor calls it (latency-bound by construction)"""
    ret, updates = _train(parl.utils.calc_gae)
    assert ret >= 150.0 and updates <= 200, (ret, updates)
