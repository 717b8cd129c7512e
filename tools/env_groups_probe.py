"""GPU box: does the emulator launch of a 512-env group (128 workgroups) keep its speed when a second group's
launch runs beside it on another stream?  (Is the lack of gain from env groups a placement collision?)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parl_amd.env import DeviceVectorEnv  # noqa: E402

dev = torch.device('cuda')
game = sys.argv[1] if len(sys.argv) > 1 else 'PongNoFrameskip-v4'


def mk(E, id0):
    e = DeviceVectorEnv(game, E, dim=42, horizon=64, seed=1, env_id0=id0, device=dev)
    e.reset()
    return e, torch.zeros(E, dtype=torch.int64, device=dev)


def run(envs, streams, steps=200):
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(steps):
        for (e, a), st in zip(envs, streams):
            with torch.cuda.stream(st):
                if e.t >= e.horizon:
                    e.roll()
                e.step_async(a)
    torch.cuda.synchronize()
    return (time.time() - t0) / steps * 1e3


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
big = mk(1024, 0)
for _ in range(2):
    print('1 x 1024 envs: %.3f ms / step' % run([big], [s1]))
del big
a, b = mk(512, 0), mk(512, 512)
for _ in range(2):
    print('1 x 512 envs alone: %.3f ms / step' % run([a], [s1]))
    print('2 x 512 envs on two streams: %.3f ms / step (both)' % run([a, b], [s1, s2]))
    print('2 x 512 envs on ONE stream: %.3f ms / step (both)' % run([a, b], [s1, s1]))
