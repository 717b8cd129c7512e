"""Dev tool (GPU box): conv12_u8_mfma_kernel against the number of observations — start-up cost of a workgroup
(weights into registers, LDS zero fill) against the per-observation cost.  Usage: python tools/conv12_scaling.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parl_amd import ops  # noqa: E402
from parl_amd.models import AtariModel42  # noqa: E402


def ev(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in e:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in e)
    return t[len(t) // 2] * 1e3


if __name__ == '__main__':
    dev = torch.device('cuda:0')
    m = AtariModel42(6).to(dev)
    for n in (128, 256, 512, 1024, 1536, 2048, 4096, 8192):
        obs = torch.randint(0, 256, (n, 4, 42, 42), dtype=torch.uint8, device=dev)
        with torch.no_grad():
            us = ev(lambda: ops.atari42_conv12(obs, m.conv1.weight, m.conv1.bias, m.conv2.weight, m.conv2.bias))
            pk = ops.atari42_conv12_pack(m.conv1.weight, m.conv2.weight)
            up = ev(lambda: ops.atari42_conv12(obs, m.conv1.weight, m.conv1.bias, m.conv2.weight, m.conv2.bias, packed=pk))
            tp = ev(lambda: ops.atari42_conv12_pack(m.conv1.weight, m.conv2.weight, out=pk))
        print('n_obs %5d: %7.1f us, operand-order weights %7.1f us (+ %.1f us to pack them)  (%.1f obs per workgroup of a %d grid)' %
              (n, us, up, tp, n / min(n, 512), min(n, 512)))
