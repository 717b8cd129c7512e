"""Dev tool (GPU box): the EFFECTIVE shader clock beside synthetic neighbours / the learner's conv kernels.  A probe
kernel (one wave per workgroup, a fixed dependent VALU chain) reads s_memtime (shader ticks) and s_memrealtime (100 MHz)
around its chain on a high-priority stream while another stream loops over the neighbour: ticks / realtime = the clock
the chain really ran at, ticks / iteration = the issue share it got.  Usage: python tools/clock_beside.py"""
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
nb.clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
sink = torch.zeros(1024, device=dev)
GRID, ITERS = 2048, 400     # 2048 probe waves (two per SIMD, like the emulator's), 400 x 64 dependent FMAs each (~100 k clocks)
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


def probe(name, fill):
    torch.cuda.synchronize()
    if fill is not None:
        with torch.cuda.stream(sb), torch.no_grad():
            fill()
    res = []
    with torch.cuda.stream(sa):
        for _ in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            nb.clock_probe_launch(out.data_ptr(), ITERS, GRID, sink.data_ptr(), sa.cuda_stream)
            b.record()
            sa.synchronize()
            o = out.view(GRID, 3).cpu()
            ticks, real = o[:, 0].double(), o[:, 1].double()
            res.append((a.elapsed_time(b) * 1e3, float(ticks.median()), float(real.median()),
                        float((ticks / real).median()) * 100.0))
    busy = not sb.query()
    torch.cuda.synchronize()
    res = res[1:]
    us = sorted(r[0] for r in res)[len(res) // 2]
    tk = sorted(r[1] for r in res)[len(res) // 2]
    mhz = sorted(r[3] for r in res)[len(res) // 2]
    print('%-62s probe launch %7.1f us | chain %8.0f shader ticks (%.2f per FMA) | ticks / realtime = %6.0f MHz%s' %
          (name, us, tk, tk / (ITERS * 64.0), mhz, '' if (fill is None or busy) else '  [neighbour ran dry]'), flush=True)


def nbr(mode):
    return lambda: [nb.neighbour_launch(mode, sink.data_ptr(), NB_ITERS.get(mode, 60), 512, sb.cuda_stream) for _ in range(300)]


probe('alone', None)
for m, nm in ((1, 'dense MFMA chains'), (2, 'VALU FMAs'), (8, 'LDS gather + 2 MFMAs per step'), (17, 'sparse MFMAs, s_sleep between pairs'),
              (28, 'as 17, AGPR accumulators'), (19, 'bursts of 16 MFMAs, long sleeps'), (26, 'tiles: gathers, wait, 32 MFMAs'),
              (29, 'as 26, AGPR accumulators'), (12, 'LDS gather + 2 VALU FMAs per step'), (32, 'as 17, random start delays'), (33, 'as 26, random start delays')):
    probe('beside neighbour %d (%s)' % (m, nm), nbr(m))
probe('beside conv12 forward, 1000 rows', lambda: [ops.atari42_conv12(obs, w1, b1, w2, b2, packed=pk) for _ in range(3000)])
probe('beside conv12 backward, 1000 rows', lambda: [ops.atari42_conv12_backward(obs, w1, b1, w2, a2, dy, packed=pk) for _ in range(2000)])
probe('beside trunk GEMM [1000,3872]x[3872,256]', lambda: [torch.mm(a2, W3.t()) for _ in range(3000)])
probe('alone again', None)
