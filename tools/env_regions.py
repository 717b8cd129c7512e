"""Dev tool (GPU box): where one launch of the env kernel goes, per env — diagnostic build with
-DPARLHIP_ENV_REGIONS (tools/build_variant.sh regions -- -DPARLHIP_ENV_REGIONS; PARL_HIP_LIB=build_exp/regions.so).
Wave A (6507 / RIOT / wrappers): s_memtime clocks in the translated cartridge code, in the interpreter's step(),
waiting for wave B (ring full, collision-latch SYNC, snapshot restore), and how often each was entered; wave B (the
picture): busy clocks (replaying records) and its lifetime.  The clock reads perturb the kernel by a few percent; the
SPLIT is what it is for."""
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
    warm = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    env = DeviceVectorEnv(game, E, dim=42, horizon=64, seed=1)
    env.reset()
    f = N.lib().parlhip_debug_env_regions
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_int]
    buf = np.zeros((E, 24), np.uint64)
    g = torch.Generator(device='cpu').manual_seed(0)
    acc = []
    for i in range(warm + 30):
        if env.t >= env.horizon:
            env.roll()
        env.step_async(torch.randint(0, env.act_dim, (E, ), generator=g).to(env.device))
        if i >= warm:
            torch.cuda.synchronize()
            assert f(buf.ctypes.data, E) == 0
            b = buf.astype(np.float64).copy()
            t = buf[:, 15]
            parts = np.stack([(t >> np.uint64(21 * k)) & np.uint64(0x1fffff) for k in range(3)], axis=1).astype(np.float64)
            b[:, 15] = parts.sum(axis=1)
            # (absolute clocks of one CU's counter) wave A's LA_EXIT -> wave B takes the batch that holds it -> wave B's fin
            d = np.stack([buf[:, 17].astype(np.int64) - buf[:, 16].astype(np.int64), buf[:, 19].astype(np.int64) - buf[:, 17].astype(np.int64),
                          buf[:, 18].astype(np.int64)], axis=1).astype(np.float64)
            acc.append(np.concatenate([b, parts, d], axis=1))
    a = np.mean(acc, axis=0)  # [E, 16] mean over launches
    m = a.mean(axis=0)
    frames = max(m[7], 1e-9)
    print('%s E=%d: per env and launch (%.1f frames): wave A %.0f clocks, wave B %.0f clocks' % (game, E, frames, m[8], m[10]))
    for i, name in ((0, 'A: translated code'), (1, 'A: interpreter step'), (2, 'A: waiting for wave B')):
        print('  %-28s %9.0f clocks = %4.1f %% of wave A, %6.1f entries per frame, %6.0f clocks per entry' %
              (name, m[i], 100 * m[i] / m[8], m[4 + i] / frames, m[i] / max(m[4 + i], 1)))
    print('  %-28s %9.0f clocks = %4.1f %% of wave A' % ('A: rest (frame loop, wrapper)', m[8] - m[0] - m[1] - m[2], 100 * (m[8] - m[0] - m[1] - m[2]) / m[8]))
    print('  %-28s %9.0f clocks before the first frame (staging, load, policy head), %.0f after the last (store %.0f, wait for the picture %.0f, observation %.0f)' % ('A: of the rest', m[3], m[15], m[24], m[25], m[26]))
    print('  %-28s wave A pushes LA_EXIT -> %.0f clocks -> wave B takes the batch that holds it (%.1f records) -> %.0f clocks -> fin' % ('the exit hand-shake:', m[27], m[29], m[28]))
    print('  %-28s %9.0f clocks = %4.1f %% of wave B\'s lifetime (the rest: polling an empty ring)' % ('B: replaying records', m[9], 100 * m[9] / max(m[10], 1)))
    print('  %-28s %9.0f records per frame, %.0f clocks per record outside tia_update' % ('B:', m[20] / frames, (m[9] - m[11]) / max(m[20], 1)))
    print('  %-28s %9.0f clocks in tia_update (%.1f calls per frame, %.0f clocks each), of it render_seg %.0f clocks (%.1f calls per frame, %.0f each)' %
          ('B:', m[11], m[13] / frames, m[11] / max(m[13], 1), m[12], m[14] / frames, m[12] / max(m[14], 1)))
