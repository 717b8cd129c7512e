"""Dev tool (GPU box): the env launch beside synthetic neighbours that stress ONE resource each (tools/neighbour_kernels.hip,
built into build_exp/neighbours.so): LDS reads, MFMA chains, VALU FMAs, or mere residency (LDS + registers held, asleep)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from parl_amd.env import DeviceVectorEnv  # noqa: E402

dev = torch.device('cuda')
nb = ctypes.CDLL(os.path.join(ROOT, 'build_exp', 'neighbours.so'))
nb.neighbour_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
E = 1024
env = DeviceVectorEnv('PongNoFrameskip-v4', E, dim=42, horizon=64, seed=1, device=dev)
env.reset()
act = torch.zeros(E, dtype=torch.int64, device=dev)
rew, don = torch.zeros(E, device=dev), torch.zeros(E, dtype=torch.uint8, device=dev)
for _ in range(30):
    env.step_async(act, rew, don)
env.roll()
sink = torch.zeros(1024, device=dev)
sa, sb = torch.cuda.Stream(priority=-1), torch.cuda.Stream()
NAMES = {0: 'LDS reads (ds_read2 gathers, no MFMA)', 1: 'MFMA chains (no LDS reads)', 2: 'VALU FMAs', 3: 'resident and asleep (LDS + registers held)',
         4: 'VALU FMAs over 200 live VGPRs', 5: '200 VGPRs held, asleep', 6: '16 KB of straight-line VALU code', 7: '48 KB of straight-line VALU code'}


def run(mode, grid=512, steps=40):
    torch.cuda.synchronize()
    if mode is not None:
        iters = {0: 400, 1: 400, 2: 400, 3: 30, 4: 40, 5: 30, 6: 6, 7: 2}[mode]
        for _ in range(300):   # ~30-100 us each: like the learner's kernels
            nb.neighbour_launch(mode, sink.data_ptr(), iters, grid, sb.cuda_stream)
    with torch.cuda.stream(sa):
        evs = []
        for i in range(steps):
            if env.t >= env.horizon:
                env.roll()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            env.step_async(act, rew, don)
            e.record()
            evs.append((s, e))
        sa.synchronize()
    busy = not sb.query()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs[5:])
    print('%-50s env launch median %.1f us (p10 %.1f, p90 %.1f)%s' % (
        ('beside %s, grid %d' % (NAMES[mode], grid)) if mode is not None else 'alone', ts[len(ts) // 2] * 1e3,
        ts[len(ts) // 10] * 1e3, ts[len(ts) * 9 // 10] * 1e3, '' if (mode is None or busy) else '  [neighbour ran dry]'))


# one neighbour launch alone, for scale
for m in range(8):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    nb.neighbour_launch(m, sink.data_ptr(), {0: 400, 1: 400, 2: 400, 3: 30, 4: 40, 5: 30, 6: 6, 7: 2}[m], 512, sb.cuda_stream)
    torch.cuda.synchronize()
    with torch.cuda.stream(sb):
        a.record()
    nb.neighbour_launch(m, sink.data_ptr(), {0: 400, 1: 400, 2: 400, 3: 30, 4: 40, 5: 30, 6: 6, 7: 2}[m], 512, sb.cuda_stream)
    with torch.cuda.stream(sb):
        b.record()
    torch.cuda.synchronize()
    print('neighbour %d (%s): one launch %.0f us' % (m, NAMES[m], a.elapsed_time(b) * 1e3))
run(None)
for m in range(8):
    run(m)
run(0, grid=256)
run(1, grid=256)
