"""Dev tool (GPU box): where an actor step goes — host enqueue time vs GPU time of the rollout's per-step chain
(stack gather -> policy forward -> sample -> emulator -> frame_post -> stats), and the chain without the emulator.
    python tools/actor_chain_probe.py [E=1024] [dim=42]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parl_amd import ops  # noqa: E402
from parl_amd.env import DeviceVectorEnv  # noqa: E402
from parl_amd.models import AtariModel42, AtariModel84  # noqa: E402
from parl_amd.rollout import DeviceRollout  # noqa: E402

if __name__ == '__main__':
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    dim = int(sys.argv[2]) if len(sys.argv) > 2 else 42
    T = 50
    dev = torch.device('cuda:0')
    env = DeviceVectorEnv('PongNoFrameskip-v4', E, dim=dim, horizon=T, seed=1, device=dev)
    model = (AtariModel42 if dim == 42 else AtariModel84)(env.act_dim).to(dev)
    for p in model.parameters():
        p.requires_grad_(False)
    ro = DeviceRollout(env, T, seed=2)
    for _ in range(3):
        ro.collect(model)
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.time()
        ro.collect_begin()
        ro.collect_steps(model)
        t1 = time.time()
        torch.cuda.synchronize()
        t2 = time.time()
        ro.collect_end()
        torch.cuda.synchronize()
        print('rollout %d: host enqueue %.1f ms (%.0f us/step), to completion %.1f ms (%.0f us/step)' %
              (rep, (t1 - t0) * 1e3, (t1 - t0) / T * 1e6, (t2 - t0) * 1e3, (t2 - t0) / T * 1e6))
    # the rollout as ONE hipGraph (what AsyncActorLearner replays between weight refreshes), nothing beside it
    if ro.can_graph(model):
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            ts = []
            for rep in range(2 + 5):   # eager, capture, then replays
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ro.collect_begin()
                a.record(side)
                ro.collect_segment(model, 0, T, graph=True)
                b.record(side)
                ro.collect_end()
                ts.append((a, b))
            side.synchronize()
            ms = sorted(a.elapsed_time(b) for a, b in ts[2:])
            print('rollout as one hipGraph, alone: %.2f ms per %d steps (%.0f us/step, %.2f M frames/s)' %
                  (ms[len(ms) // 2], T, ms[len(ms) // 2] / T * 1e3, 4 * T * E / ms[len(ms) // 2] / 1e3))
        torch.cuda.current_stream(dev).wait_stream(side)
    # the chain without the emulator: forward + sample only
    obs = env.current_obs()
    logits = torch.zeros((E, env.act_dim), device=dev)
    act = torch.zeros(E, dtype=torch.int64, device=dev)

    def chain():
        with torch.no_grad():
            o = env.current_obs(ro._obs_step)
            if hasattr(model, 'policy_into'):
                model.policy_into(o, logits)
            else:
                logits.copy_(model.policy(o))
            ops.policy_sample_into(logits, act, 1, 5, 0)

    for _ in range(5):
        chain()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(50):
        chain()
    t1 = time.time()
    torch.cuda.synchronize()
    t2 = time.time()
    print('policy chain alone: host %.0f us/step, GPU-bound %.0f us/step' % ((t1 - t0) / 50 * 1e6, (t2 - t0) / 50 * 1e6))

    def envonly():
        env.step_async(act, ro.rewards[0], ro.dones[0])
        env.accumulate_episode_stats(ro.ep_stats)

    env.roll()
    for _ in range(3):
        envonly()
    torch.cuda.synchronize()
    env.roll()
    t0 = time.time()
    for _ in range(40):
        envonly()
    t1 = time.time()
    torch.cuda.synchronize()
    t2 = time.time()
    print('env step + frame_post + stats alone: host %.0f us/step, GPU-bound %.0f us/step' %
          ((t1 - t0) / 40 * 1e6, (t2 - t0) / 40 * 1e6))
