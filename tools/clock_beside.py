"""Dev tool (GPU box): which INSTRUCTION CLASS of a latency-bound wave suffers beside a neighbour, and the effective
shader clock meanwhile.  Probe kernels (tools/neighbour_kernels.hip: one wave per workgroup, a fixed dependent chain of
VALU FMAs / SALU adds / LDS pointer chasing / scalar-cache pointer chasing / VALU + branches) read s_memtime (shader
ticks) and s_memrealtime (100 MHz) around their chain on a high-priority stream while another stream loops over the
neighbour: ticks / realtime = the clock the chain really ran at, ticks / step = what the neighbour costs that class.
Probes run at s_setprio 0 and 3 (the emulator's).  Usage: python tools/clock_beside.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from parl_amd import ops  # noqa: E402

dev = torch.device('cuda')
nb = ctypes.CDLL(os.path.join(ROOT, 'build_exp', 'neighbours.so'))
nb.neighbour_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
nb.probe_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                            ctypes.c_void_p, ctypes.c_void_p]
sink = torch.zeros(1024, device=dev)
GRID = 2048                 # 2048 probe waves: two per SIMD, like the emulator's
# (kind, name, iterations, steps per iteration)
KINDS = [(0, 'VALU FMA chain', 400, 64), (1, 'SALU chain', 400, 128), (2, 'LDS pointer chase', 100, 16),
         (3, 'scalar-cache pointer chase', 100, 8), (4, 'VALU + branch', 400, 32)]
chase = (((torch.arange(1024, dtype=torch.int32, device=dev) * 17 + 61) % 1024) * 4).contiguous()   # byte offsets: a 1024-element cycle (4 KB)
out = torch.zeros(3 * GRID, dtype=torch.int64, device=dev)
sa, sb = torch.cuda.Stream(priority=-1), torch.cuda.Stream()
NB_ITERS = {1: 400, 2: 400, 22: 200, 23: 100, 24: 100, 25: 100, 26: 100, 29: 100, 31: 100, 33: 100}
R = 1000
obs = torch.randint(0, 256, (R, 4, 42, 42), dtype=torch.uint8, device=dev)
w1, b1 = torch.randn(16, 4, 4, 4, device=dev) * 0.2, torch.zeros(16, device=dev)
w2, b2 = torch.randn(32, 16, 4, 4, device=dev) * 0.1, torch.zeros(32, device=dev)
pk = ops.atari42_conv12_pack(w1, w2)
a2 = ops.atari42_conv12(obs, w1, b1, w2, b2, packed=pk)
dy = torch.randn_like(a2)
W3 = torch.randn(256, 3872, device=dev) * 0.01
BASE = {}


def probe(name, fill):
    torch.cuda.synchronize()
    if fill is not None:
        with torch.cuda.stream(sb), torch.no_grad():
            fill()
    cells, mhz_all = [], []
    with torch.cuda.stream(sa):
        for prio in (0, 3):
            for kind, kname, iters, per in KINDS:
                vals = []
                for rep in range(4):
                    nb.probe_launch(kind, prio, out.data_ptr(), iters, GRID, sink.data_ptr(), chase.data_ptr(), sa.cuda_stream)
                    sa.synchronize()
                    o = out.view(GRID, 3).cpu()
                    ticks, real = o[:, 0].double(), o[:, 1].double()
                    if rep:
                        vals.append(float(ticks.median()) / (iters * per))
                        mhz_all.append(float((ticks / real).median()) * 100.0)
                v = sorted(vals)[len(vals) // 2]
                key = (prio, kind)
                if fill is None and key not in BASE:
                    BASE[key] = v
                cells.append('%6.1f (x%.2f)' % (v, v / BASE[key]))
    busy = not sb.query()
    torch.cuda.synchronize()
    mhz = sorted(mhz_all)[len(mhz_all) // 2]
    print('%-46s | prio 0: %s | prio 3: %s | %4.0f MHz%s' % (name, ' '.join(cells[:5]), ' '.join(cells[5:]), mhz,
                                                           '' if (fill is None or busy) else '  [neighbour ran dry]'), flush=True)


def nbr(mode):
    return lambda: [nb.neighbour_launch(mode, sink.data_ptr(), NB_ITERS.get(mode, 60), 512, sb.cuda_stream) for _ in range(1200)]


print('shader ticks per step of a dependent chain (x = against the same probe alone); columns per priority: ' +
      ', '.join(k[1] for k in KINDS))
probe('alone', None)
for m, nm in ((1, 'dense MFMA chains'), (2, 'VALU FMAs'), (0, 'LDS gathers, no MFMA'), (8, 'LDS gather + 2 MFMAs per step'),
              (17, 'sparse MFMAs, s_sleep between pairs'), (28, 'as 17, AGPR accumulators'), (26, 'tiles: gathers, wait, 32 MFMAs'),
              (12, 'LDS gather + 2 VALU FMAs per step'), (13, 'LDS gathers + independent dense MFMAs')):
    probe('beside %d (%s)' % (m, nm), nbr(m))
probe('beside conv12 forward, 1000 rows', lambda: [ops.atari42_conv12(obs, w1, b1, w2, b2, packed=pk) for _ in range(6000)])
probe('beside conv12 backward, 1000 rows', lambda: [ops.atari42_conv12_backward(obs, w1, b1, w2, a2, dy, packed=pk) for _ in range(4000)])
probe('beside trunk GEMM [1000,3872]x[3872,256]', lambda: [torch.mm(a2, W3.t()) for _ in range(8000)])
probe('alone again', None)
