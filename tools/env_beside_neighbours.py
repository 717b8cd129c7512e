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
def sc_busy():
    return 'sc' in globals() and not sc.query()


NAMES = {0: 'LDS reads (ds_read2 gathers, no MFMA)', 1: 'MFMA chains (no LDS reads)', 2: 'VALU FMAs', 3: 'resident and asleep (LDS + registers held)',
         4: 'VALU FMAs over 200 live VGPRs', 5: '200 VGPRs held, asleep', 6: '16 KB of straight-line VALU code', 7: '48 KB of straight-line VALU code', 8: 'LDS gather + 2 MFMAs per step, 64 operand registers, noisy data',
         9: 'as 8 without the LDS gather in the loop', 10: 'as 8 with quiet data (all ones)', 11: 'as 8 with one operand register pair',
         12: 'LDS gather + 2 VALU FMAs per step (no MFMA)', 13: 'LDS gathers and MFMAs interleaved, independent',
         14: 'as 8, the gather two steps ahead of its MFMAs', 15: 'as 8, gathered value through v_mov_b32 first',
         16: 'as 8, gathered value through v_add_f32 0 first', 17: 'sparse MFMAs, no LDS: s_sleep between pairs',
         18: 'sparse MFMAs, no LDS: a VALU chain between pairs', 19: 'MFMAs in bursts of 16 with long sleeps, no LDS',
         20: 'sparse 4x4x1 MFMAs (2 passes), s_sleep between pairs', 21: 'sparse 32x32x2 MFMAs (16 passes), s_sleep between pairs',
         22: 'ONE accumulator: dense dependent MFMA chain, no LDS', 23: 'software-pipelined tiles: gathers of tile t+1 before the 32 MFMAs of tile t, epilogue',
         24: 'as 23 + a workgroup barrier every 4 tiles', 25: 'as 23 without the epilogue', 26: 'as 23 NOT pipelined (gathers waited for in front of their MFMAs)',
         27: 'as 8, accumulators in AGPRs', 28: 'as 17, accumulators in AGPRs', 29: 'as 26, accumulators in AGPRs (one pair across tiles)',
         30: 'as 8 at s_setprio 3', 31: 'as 26 at s_setprio 3 (one accumulator pair across tiles)',
         32: 'as 17, every wave starts after a pseudo-random delay', 33: 'as 26 (one accumulator pair), every wave starts after a pseudo-random delay'}
ITERS = {0: 400, 1: 400, 2: 400, 3: 30, 4: 40, 5: 30, 6: 6, 7: 2, 22: 200, 23: 100, 24: 100, 25: 100, 26: 100, 29: 100, 31: 100, 33: 100}
MODES = [int(x) for x in os.environ['NEIGHBOUR_MODES'].split(',')] if os.environ.get('NEIGHBOUR_MODES') else list(range(34))


def run(mode, grid=512, steps=40):
    if not sc_busy():
        torch.cuda.synchronize()
    if mode is not None:
        iters = ITERS.get(mode, 60)
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
for m in MODES:
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    nb.neighbour_launch(m, sink.data_ptr(), ITERS.get(m, 60), 512, sb.cuda_stream)
    torch.cuda.synchronize()
    with torch.cuda.stream(sb):
        a.record()
    nb.neighbour_launch(m, sink.data_ptr(), ITERS.get(m, 60), 512, sb.cuda_stream)
    with torch.cuda.stream(sb):
        b.record()
    torch.cuda.synchronize()
    print('neighbour %d (%s): one launch %.0f us' % (m, NAMES[m], a.elapsed_time(b) * 1e3))
run(None)
for m in MODES:
    run(m)
if not os.environ.get('NEIGHBOUR_MODES'):
    run(0, grid=256)
    run(1, grid=256)

# a "heater": dense MFMAs on a THIRD stream (one workgroup per CU) beside the sparse-MFMA neighbour — does a matrix pipe that
# never goes idle take the stretch away?
sc = torch.cuda.Stream()


def run_heated(mode, heater_iters=120000):
    torch.cuda.synchronize()
    nb.neighbour_launch(1, sink.data_ptr(), heater_iters, 256, sc.cuda_stream)
    run(mode)


if os.environ.get('HEATER'):
    print('--- with a dense-MFMA heater kernel (grid 256) on a third stream')
    run_heated(None)
    run_heated(17)
    run_heated(8)
