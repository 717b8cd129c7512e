"""Dev tool (GPU box): raw env-step throughput of the device emulator (random actions)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parl_amd.env import DeviceVectorEnv  # noqa: E402

if __name__ == '__main__':
    game = sys.argv[1] if len(sys.argv) > 1 else 'PongNoFrameskip-v4'
    for E in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else '256,1024,4096').split(',')]:
        env = DeviceVectorEnv(game, E, dim=84, horizon=64, seed=1)
        env.reset()
        torch.cuda.synchronize()
        acts = torch.randint(0, env.act_dim, (64, E), device=env.device)
        for warm in (True, False):
            env.roll()
            t0 = time.time()
            n = 8 if warm else 48
            for i in range(n):
                env.step_async(acts[i])
            torch.cuda.synchronize()
            dt = time.time() - t0
        print('%s E=%d: %.2f ms/step  %.0f agent-steps/s  %.0f emulated frames/s  jam=%x' %
              (game, E, dt / n * 1e3, n * E / dt, 4 * n * E / dt, int(env.jam.item())))
