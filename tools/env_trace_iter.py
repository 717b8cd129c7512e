"""Dev tool (GPU box): head-to-head clocks of ONE translated loop trace, as a histogram — diagnostic build with
-DPARLHIP_ENV_REGIONS -DPARLHIP_ENV_TRACEITER=0xf5e0 (tools/build_variant.sh traceiter -- -DPARLHIP_ENV_REGIONS
-DPARLHIP_ENV_TRACEITER=0xf5e0; PARL_HIP_LIB=build_exp/traceiter.so).  Says what one iteration of the cartridge's hot
loop costs on the device when nothing interrupts it, and how many iterations are interrupted."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parl_amd import _native as N  # noqa: E402
from parl_amd.env import DeviceVectorEnv  # noqa: E402

if __name__ == '__main__':
    game = sys.argv[1] if len(sys.argv) > 1 else 'PongNoFrameskip-v4'
    E = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    warm = 40
    env = DeviceVectorEnv(game, E, dim=42, horizon=64, seed=1)
    env.reset()
    f = N.lib().parlhip_debug_env_regions
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_int]
    buf = np.zeros((E, 16), np.uint64)
    g = torch.Generator(device='cpu').manual_seed(0)
    acc = []
    for i in range(warm + 30):
        if env.t >= env.horizon:
            env.roll()
        env.step_async(torch.randint(0, env.act_dim, (E, ), generator=g).to(env.device))
        if i >= warm:
            torch.cuda.synchronize()
            assert f(buf.ctypes.data, -E) == 0
            acc.append(buf.astype(np.float64).copy())
    m = np.mean(acc, axis=0).mean(axis=0)
    print('%s E=%d: per env and launch: %.1f uninterrupted passes of the trace head, %.0f clocks each; %.1f other passes' %
          (game, E, m[1], m[0] / max(m[1], 1e-9), m[2]))
