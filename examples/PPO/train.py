"""examples/PPO/train.py of the reference on the device path (same loop, train.py:45-129).

    python examples/PPO/train.py --env PongNoFrameskip-v4 --env_num 8            # Atari, envs on the GPU
    python examples/PPO/train.py --continuous_action --env_num 64                # host-stepped sims + GPU path
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from parl_amd.algorithms import PPO  # noqa: E402
from parl_amd.utils import logger  # noqa: E402

from agent import PPOAgent  # noqa: E402
from atari_config import atari_config  # noqa: E402
from atari_model import AtariModel  # noqa: E402
from env_utils import ParallelEnv  # noqa: E402
from mujoco_config import mujoco_config  # noqa: E402
from mujoco_model import MujocoModel  # noqa: E402
from parl_amd.storage import RolloutStorage  # noqa: E402  (device twin of examples/PPO/storage.py)


def main(args):
    config = dict(mujoco_config if args.continuous_action else atari_config)
    if args.env_num:
        config['env_num'] = args.env_num
    if args.step_nums:
        config['step_nums'] = args.step_nums
    if not args.continuous_action:
        config['env'] = args.env
    config['seed'] = args.seed
    config['train_total_steps'] = int(args.train_total_steps)
    config['batch_size'] = int(config['env_num'] * config['step_nums'])
    config['num_updates'] = max(1, int(config['train_total_steps'] // config['batch_size']))
    dev = torch.device('cuda:0')
    envs = ParallelEnv(config, device=dev)
    model = (MujocoModel if config['continuous_action'] else AtariModel)(envs.obs_space, envs.act_space)
    ppo = PPO(model, clip_param=config['clip_param'], entropy_coef=config['entropy_coef'],
              initial_lr=config['initial_lr'], continuous_action=config['continuous_action'])
    agent = PPOAgent(ppo, config)
    rollout = RolloutStorage(config['step_nums'], config['env_num'], envs.obs_space, envs.act_space, device=dev)
    obs = envs.reset()
    done = torch.zeros(config['env_num'], device=dev)
    total_steps, t0 = 0, time.time()
    for update in range(1, config['num_updates'] + 1):
        for step in range(config['step_nums']):
            total_steps += config['env_num']
            value, action, logprob, _ = agent.sample(obs)
            next_obs, reward, next_done = envs.step(action)
            rollout.append(obs, action, logprob, reward, done, value.flatten())
            obs, done = next_obs, next_done
        value = agent.value(obs)                      # bootstrap value if not done (train.py:105-106)
        rollout.compute_returns(value.flatten(), done)
        value_loss, action_loss, entropy_loss, lr = agent.learn(rollout)
        if update % args.log_interval == 0 or update == config['num_updates']:
            n, mean_r, mean_l = envs.pop_episode_stats()
            logger.info({'update': update, 'total_steps': total_steps, 'value_loss': value_loss,
                         'action_loss': action_loss, 'entropy_loss': entropy_loss, 'lr': lr, 'episodes': n,
                         'mean_episode_reward': mean_r, 'steps_per_sec': total_steps / (time.time() - t0)})


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('--env', type=str, default='PongNoFrameskip-v4')
    parser.add_argument('--seed', type=int, default=None)
    parser.add_argument('--env_num', type=int, default=None)
    parser.add_argument('--step_nums', type=int, default=None)
    parser.add_argument('--continuous_action', action='store_true', default=False)
    parser.add_argument('--train_total_steps', type=float, default=10e6)
    parser.add_argument('--log-interval', type=int, default=1)
    main(parser.parse_args())
