"""Dev tool (GPU box): per-WAVE duration of the env kernel inside one launch (diagnostic build with
-DPARLHIP_ENV_TIMING: tools/build_exp.sh timing:"-DPARLHIP_ENV_TIMING"; PARL_HIP_LIB=build_exp/timing.so).
A launch lasts as long as its slowest wave: how far is that from the mean wave?"""
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
    warm = int(sys.argv[3]) if len(sys.argv) > 3 else 600
    env = DeviceVectorEnv(game, E, dim=42, horizon=64, seed=1)
    env.reset()
    L = N.lib()
    f = L.parlhip_debug_env_clocks
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    t0, t1 = np.zeros(E, np.uint64), np.zeros(E, np.uint64)
    g = torch.Generator(device='cpu').manual_seed(0)
    rows = []
    for i in range(warm + 100):
        if env.t >= env.horizon:
            env.roll()
        env.step_async(torch.randint(0, env.act_dim, (E, ), generator=g).to(env.device))
        if i >= warm or i < 20:
            torch.cuda.synchronize()
            assert f(t0.ctypes.data, t1.ctypes.data, E) == 0
            d = (t1 - t0).astype(np.float64) / 100.0  # wall_clock64: 100 MHz -> us
            span = (t1.max() - t0.min()) / 100.0
            rows.append((i, d.mean(), np.percentile(d, 50), np.percentile(d, 90), np.percentile(d, 99), d.max(), span))
    for lo, hi, name in ((0, 20, 'first 20 steps'), (20, len(rows), 'steps %d..%d' % (warm, warm + 100))):
        a = np.array(rows[lo:hi])
        print('%s E=%d %s: wave us mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f | launch span %.0f | max/mean %.2f' %
              (game, E, name, a[:, 1].mean(), a[:, 2].mean(), a[:, 3].mean(), a[:, 4].mean(), a[:, 5].mean(), a[:, 6].mean(),
               a[:, 5].mean() / a[:, 1].mean()))
