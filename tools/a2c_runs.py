"""Dev tool (GPU box): the first N seconds of examples/A2C/train.py under several seeds / schedules in one process:
vf_loss (window mean), entropy, the share of fc units that are zero on a whole batch (dead ReLUs) and the largest
|value| — does the value function start to fit (profiles/r03_a2c_*: 424 -> 200 within 4e6 sample steps)?
usage: a2c_runs.py SECONDS 'seed:max_sample_steps' ..."""
import copy
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'examples', 'A2C'))
sys.path.insert(0, ROOT)
import train as a2c_train  # noqa: E402
from a2c_config import config as base  # noqa: E402


def run(seconds, seed, max_steps, adam_eps=None):
    cfg = copy.deepcopy(base)
    cfg['max_sample_steps'] = max_steps
    if seed >= 0:
        cfg['seed'] = seed
        torch.manual_seed(seed)
    ln = a2c_train.Learner(cfg)
    if adam_eps is not None:
        for g in ln.agent.alg.optimizer.param_groups:
            g['eps'] = adam_eps
    model = ln.agent.alg.model
    t0, nxt = time.time(), 5.0
    while time.time() - t0 < seconds:
        ln.step()
        if time.time() - t0 >= nxt or ln.sample_total_steps <= 5120 * 12:
            ro = ln.remote_actors[0].rollout
            with torch.no_grad():
                h = model._trunk(ro.obs[:2560])
                v = model.value_fc(h)
            print('  seed %d steps %9d t %5.1f  vf %8.2f ent %7.1f  dead fc %.3f  |v|max %.3f  lr %.3g' %
                  (seed, ln.sample_total_steps, time.time() - t0, ln.vf_loss_stat.mean, ln.entropy_stat.mean,
                   float((h.max(0)[0] <= 0).float().mean()), float(v.abs().max()), ln.lr), flush=True)
            if time.time() - t0 >= nxt:
                nxt += 5.0


if __name__ == '__main__':
    secs = float(sys.argv[1])
    for spec in sys.argv[2:]:
        parts = spec.split(':')
        seed, ms = int(parts[0]), int(float(parts[1]))
        eps = float(parts[2]) if len(parts) > 2 else None
        print('run seed=%d max_sample_steps=%d eps=%s' % (seed, ms, eps), flush=True)
        run(secs, seed, ms, eps)
