"""Dev tool (GPU box): where the translated cartridge code spends wave A's clocks, by the PC it was ENTERED at —
diagnostic build with -DPARLHIP_ENV_REGIONS -DPARLHIP_ENV_ENTRYHIST (tools/build_variant.sh entryhist --
-DPARLHIP_ENV_REGIONS -DPARLHIP_ENV_ENTRYHIST; PARL_HIP_LIB=build_exp/entryhist.so).  One row per dispatch entry:
entries per frame, 6507 instructions and clocks per entry, clocks per 6507 instruction."""
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
    warm, steps = 40, 30
    env = DeviceVectorEnv(game, E, dim=42, horizon=64, seed=1)
    env.reset()
    f = N.lib().parlhip_debug_env_regions
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_int]
    buf = np.zeros((8192, 3), np.uint64)
    reg = np.zeros((E, 16), np.uint64)
    g = torch.Generator(device='cpu').manual_seed(0)
    frames = 0.0
    for i in range(warm + steps):
        if env.t >= env.horizon:
            env.roll()
        if i == warm:
            torch.cuda.synchronize()
            assert f(buf.ctypes.data, -8192) == 0  # clears the table
        env.step_async(torch.randint(0, env.act_dim, (E, ), generator=g).to(env.device))
        if i >= warm:
            torch.cuda.synchronize()
            assert f(reg.ctypes.data, E) == 0
            frames += reg[:, 7].astype(np.float64).sum()
    torch.cuda.synchronize()
    assert f(buf.ctypes.data, -8192) == 0
    h = buf.astype(np.float64)
    tot = h[:, 0].sum()
    print('%s E=%d: translated code by entry PC, per emulated frame (%.0f frames): %.0f clocks, %.0f 6507 instructions, %.1f entries' %
          (game, E, frames, tot / frames, h[:, 1].sum() / frames, h[:, 2].sum() / frames))
    for k in np.argsort(-h[:, 0])[:40]:
        if h[k, 2] == 0:
            break
        print('  entry %04x  %6.2f per frame  %8.0f clocks per frame (%4.1f %%)  %7.1f instr per entry  %8.0f clocks per entry  %6.1f clocks per 6507 instr' %
              (0xe000 | k, h[k, 2] / frames, h[k, 0] / frames, 100 * h[k, 0] / tot, h[k, 1] / h[k, 2], h[k, 0] / h[k, 2], h[k, 0] / max(h[k, 1], 1)))
