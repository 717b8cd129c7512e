"""Dev tool (GPU box): per-wave START and DURATION of the env kernel inside one launch, alone and beside a looping
conv12 forward (diagnostic build -DPARLHIP_ENV_TIMING: tools/build_obj_variant.sh timing atari_env.hip -DPARLHIP_ENV_TIMING;
PARL_HIP_LIB=build_exp/timing.so).  Is the stretch late starts (dispatch) or slow waves (execution)?"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parl_amd import _native as N  # noqa: E402
from parl_amd import ops  # noqa: E402
from parl_amd.env import DeviceVectorEnv  # noqa: E402

dev = torch.device('cuda')
E = 1024
env = DeviceVectorEnv('PongNoFrameskip-v4', E, dim=42, horizon=64, seed=1, device=dev)
env.reset()
f = N.lib().parlhip_debug_env_clocks
f.restype, f.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
t0, t1 = np.zeros(E, np.uint64), np.zeros(E, np.uint64)
act = torch.zeros(E, dtype=torch.int64, device=dev)
obs = torch.randint(0, 256, (1000, 4, 42, 42), dtype=torch.uint8, device=dev)
w1, b1 = torch.randn(16, 4, 4, 4, device=dev) * 0.2, torch.zeros(16, device=dev)
w2, b2 = torch.randn(32, 16, 4, 4, device=dev) * 0.1, torch.zeros(32, device=dev)
pk = ops.atari42_conv12_pack(w1, w2)
sa, sb = torch.cuda.Stream(priority=-1), torch.cuda.Stream()
for _ in range(30):
    env.step_async(act)
env.roll()
torch.cuda.synchronize()
for beside in (False, True, False, True):
    rows = []
    for i in range(12):
        if env.t >= env.horizon:
            env.roll()
        torch.cuda.synchronize()
        if beside:
            with torch.cuda.stream(sb), torch.no_grad():
                for _ in range(60):
                    ops.atari42_conv12(obs, w1, b1, w2, b2, packed=pk)
        with torch.cuda.stream(sa):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            env.step_async(act)
            b.record()
        torch.cuda.synchronize()
        assert f(t0.ctypes.data, t1.ctypes.data, E) == 0
        start = (t0 - t0.min()).astype(np.float64) / 100.0   # wall_clock64: 100 MHz -> us
        dur = (t1 - t0).astype(np.float64) / 100.0
        rows.append((a.elapsed_time(b) * 1e3, start.mean(), np.percentile(start, 90), start.max(), dur.mean(), np.percentile(dur, 90),
                     dur.max(), (t1.max() - t0.min()) / 100.0))
    r = np.array(rows[2:]).mean(0)
    print('%-22s launch (events) %.0f us | wave start after the first: mean %.0f p90 %.0f max %.0f | wave duration: mean %.0f p90 %.0f '
          'max %.0f | first start -> last end %.0f' % ('beside conv12 forward' if beside else 'alone', *r))
