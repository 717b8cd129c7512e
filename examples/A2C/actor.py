"""Actor with the interface of examples/A2C/actor.py:27-119 on the device path."""
from collections import defaultdict

import torch

import parl_amd as parl
from atari_agent import AtariAgent
from parl_amd.models import AtariModel84 as AtariModel  # torch twin of examples/A2C/atari_model.py:21-104
from parl_amd.algorithms import A2C
from parl_amd.env import DeviceVectorEnv
from parl_amd.rollout import DeviceA2CRollout


@parl.remote_class(wait=False)
class Actor(object):
    def __init__(self, config, actor_id=0, model=None, device=None):
        self.config = config
        E, T = config['env_num'], config['sample_batch_steps']
        self.vector_env = DeviceVectorEnv(config['env_name'], E, dim=config['env_dim'], horizon=T,
                                          seed=config.get('seed', 0), env_id0=actor_id * E, device=device)
        self.config['act_dim'] = self.vector_env.act_dim
        self.config['obs_shape'] = self.vector_env.obs_shape
        self.shared = model is not None
        model = model if model is not None else AtariModel(self.vector_env.act_dim)
        self.agent = AtariAgent(A2C(model, vf_loss_coeff=config['vf_loss_coeff']), config,
                                device=self.vector_env.device)
        self.rollout = DeviceA2CRollout(self.vector_env, T, config['gamma'], config['lambda'],
                                        seed=config.get('seed', 0) + 1000 + actor_id)

    def sample(self):
        batch = self.rollout.collect(self.agent.alg.model)
        torch.cuda.current_stream().synchronize()
        return batch

    def get_metrics(self):
        metrics = defaultdict(list)
        n, mean_r, mean_l = self.rollout.pop_episode_stats()
        if n:
            metrics['episode_rewards'] += [mean_r] * int(n)
            metrics['episode_steps'] += [mean_l] * int(n)
        return metrics

    def set_weights(self, params):
        if not self.shared:
            self.agent.set_weights(params)
