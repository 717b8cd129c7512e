"""Dev tool (GPU box): per-phase wall times of one IMPALA actor-learner iteration."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import parl_amd as parl  # noqa: E402
from parl_amd.env import DeviceVectorEnv  # noqa: E402
from parl_amd.models import AtariModel42, AtariModel84  # noqa: E402
from parl_amd.rollout import DeviceRollout  # noqa: E402

T0 = time.time()


def lap(tag, t):
    torch.cuda.synchronize()
    print('[%7.2fs] %-28s %.3f s' % (time.time() - T0, tag, time.time() - t), flush=True)
    return time.time()


if __name__ == '__main__':
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    dim = int(sys.argv[2]) if len(sys.argv) > 2 else 42
    T = 50
    dev = torch.device('cuda:0')
    t = time.time()
    env = DeviceVectorEnv('PongNoFrameskip-v4', E, dim=dim, horizon=T, seed=1, device=dev)
    t = lap('env create + reset cache', t)
    model = (AtariModel42 if dim == 42 else AtariModel84)(env.act_dim).to(dev)
    alg = parl.algorithms.IMPALA(model, sample_batch_steps=T, gamma=0.99, vf_loss_coeff=0.5,
                                 clip_rho_threshold=1.0, clip_pg_rho_threshold=1.0)
    ro = DeviceRollout(env, T, seed=2)
    t = lap('model', t)
    env.reset()
    ro.started = True
    t = lap('env.reset', t)
    with torch.no_grad():
        model.policy(env.current_obs())
    t = lap('first policy fwd', t)
    for it in range(3):
        b = ro.collect(model)
        t = lap('collect %d' % it, t)
        loss, kl = alg.learn(b['obs'], b['actions'], b['behaviour_logits'], b['rewards'], b['dones'], 1e-3, -0.01,
                             time_major=True)
        t = lap('learn %d' % it, t)
    print('loss', float(loss.total_loss.item()), 'jam', int(env.jam.item()), 'episodes', ro.pop_episode_stats())
