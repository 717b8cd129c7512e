"""A2C learner — the synchronous step() of examples/A2C/train.py:30-186 (kick off all actors,
collect, one update) on the device path.

    python examples/A2C/train.py [--max_sample_steps N] [--env-num E] [--horizon updates|samples] [--seed S] [--minutes M]

The config file carries the reference's `max_sample_steps` = 1e7, which its 5 x 5 CPU envs consume as 20,000
updates of 500 rows.  An update here has env_num x actor_num x 20 rows (256 envs: 5,120), so the same number of
sample steps would be 1,953 updates — measured on one MI355X (profiles/r03_a2c_pong_256envs_*.log) that ends
at -20.2, the policy never leaves uniform, while the reference's number of UPDATES takes Pong across 0 after
1.85e7 steps (130 s) to +20.2 after 3.3e7 (230 s).  `--horizon updates` (the default) therefore scales the
config's value by (envs here) / (the reference's 25 envs) and logs it; `--horizon samples` or an explicit
`--max_sample_steps N` uses the number as it stands.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import parl_amd as parl  # noqa: E402
from actor import Actor  # noqa: E402
from atari_agent import AtariAgent  # noqa: E402
from parl_amd.models import AtariModel84 as AtariModel  # noqa: E402  (torch twin of examples/A2C/atari_model.py:21-104)
from parl_amd.algorithms import A2C  # noqa: E402
from parl_amd.env import GAMES  # noqa: E402
from parl_amd.utils import logger, summary  # noqa: E402
from parl_amd.utils.time_stat import TimeStat  # noqa: E402
from parl_amd.utils.window_stat import WindowStat  # noqa: E402


class Learner(object):
    def __init__(self, config):
        self.config = config
        self.device = torch.device('cuda')
        act_dim = 6 if GAMES[config['env_name']][0] == 'pong' else 4
        self.config['act_dim'] = act_dim
        self.config['obs_shape'] = (4, config['env_dim'], config['env_dim'])
        model = AtariModel(act_dim)
        self.agent = AtariAgent(A2C(model, vf_loss_coeff=config['vf_loss_coeff']), config, device=self.device)
        self.total_loss_stat = WindowStat(100)
        self.pi_loss_stat = WindowStat(100)
        self.vf_loss_stat = WindowStat(100)
        self.entropy_stat = WindowStat(100)
        self.lr = self.entropy_coeff = None
        self.learn_time_stat = TimeStat(100)
        self.sample_total_steps = 0
        parl.connect(config['master_address'])
        self.remote_actors = [
            Actor(config, i, model=self.agent.alg.model, device=self.device) for i in range(config['actor_num'])
        ]
        self.start_time = time.time()

    def step(self):
        latest_params = None if all(a.shared for a in self.remote_actors) else self.agent.get_weights()
        for a in self.remote_actors:
            a.set_weights(latest_params)
        futures = [a.sample() for a in self.remote_actors]
        datas = [f.get() for f in futures]
        batch = {k: (torch.cat([d[k] for d in datas]) if len(datas) > 1 else datas[0][k]) for k in datas[0]}
        self.sample_total_steps += int(batch['actions'].numel())
        with self.learn_time_stat:
            total_loss, pi_loss, vf_loss, entropy, lr, entropy_coeff = self.agent.learn(
                obs_np=batch['obs'], actions_np=batch['actions'], advantages_np=batch['advantages'],
                target_values_np=batch['target_values'])
        self.total_loss_stat.add(total_loss)
        self.pi_loss_stat.add(pi_loss)
        self.vf_loss_stat.add(vf_loss)
        self.entropy_stat.add(entropy)
        self.lr, self.entropy_coeff = lr, entropy_coeff

    def log_metrics(self):
        metrics = [a.get_metrics().get() for a in self.remote_actors]
        rewards = [r for m in metrics for r in m['episode_rewards']]
        steps = [s for m in metrics for s in m['episode_steps']]
        elapsed = time.time() - self.start_time
        metric = {
            'sample_steps': self.sample_total_steps,
            'mean_episode_rewards': float(np.mean(rewards)) if rewards else None,
            'mean_episode_steps': float(np.mean(steps)) if steps else None,
            'episodes': len(rewards),
            'total_loss': self.total_loss_stat.mean,
            'pi_loss': self.pi_loss_stat.mean,
            'vf_loss': self.vf_loss_stat.mean,
            'entropy': self.entropy_stat.mean,
            'learn_time_s': self.learn_time_stat.mean,
            'elapsed_time_s': int(elapsed),
            'env_frames_per_s': 4 * self.sample_total_steps / max(elapsed, 1e-9),
            'lr': self.lr,
            'entropy_coeff': self.entropy_coeff,
        }
        metric = {k: (float(v) if isinstance(v, np.floating) else v) for k, v in metric.items()}   # plain numbers in the log
        for key, value in metric.items():
            if value is not None:
                summary.add_scalar(key, value, self.sample_total_steps)
        logger.info(metric)
        return metric

    def should_stop(self):
        return self.sample_total_steps >= self.config['max_sample_steps']


if __name__ == '__main__':
    from a2c_config import config
    parser = argparse.ArgumentParser()
    parser.add_argument('--max_sample_steps', type=int, default=None, help='stop condition: number of sample step')
    parser.add_argument('--env-num', type=int, default=None)
    parser.add_argument('--log-interval', type=float, default=None)
    parser.add_argument('--horizon', choices=('updates', 'samples'), default='updates',
                        help='what the config\'s max_sample_steps preserves when the actor pool is bigger than the '
                        'reference\'s 25 envs: the number of updates (scaled sample steps) or the sample steps')
    parser.add_argument('--minutes', type=float, default=None, help='stop after this many minutes (the schedules keep '
                        'their horizon: a look at the beginning of the run)')
    parser.add_argument('--seed', type=int, default=None,
                        help='seed of the network initialisation, the envs and the sampler (default: envs / sampler 0, '
                        'torch\'s own default generator for the initialisation)')
    args = parser.parse_args()
    if args.seed is not None:
        config['seed'] = args.seed
        torch.manual_seed(args.seed)
        np.random.seed(args.seed)
    if args.env_num:
        config['env_num'] = args.env_num
    if args.max_sample_steps is not None:
        config['max_sample_steps'] = args.max_sample_steps
    elif args.horizon == 'updates':
        REFERENCE_ENVS = 5 * 5  # actor_num x env_num of the reference's a2c_config.py:22-26
        envs = config['env_num'] * config['actor_num']
        config['max_sample_steps'] = int(config['max_sample_steps'] * max(1.0, envs / REFERENCE_ENVS))
        logger.info('max_sample_steps scaled to %d: the reference\'s %d updates at %d rows per update' %
                    (config['max_sample_steps'], config['max_sample_steps'] // (envs * config['sample_batch_steps']),
                     envs * config['sample_batch_steps']))
    if args.log_interval:
        config['log_metrics_interval_s'] = args.log_interval
    learner = Learner(config)
    assert config['log_metrics_interval_s'] > 0
    deadline = time.time() + 60.0 * args.minutes if args.minutes else float('inf')
    while not learner.should_stop() and time.time() < deadline:
        start = time.time()
        while time.time() - start < config['log_metrics_interval_s'] and not learner.should_stop():
            learner.step()
        learner.log_metrics()
