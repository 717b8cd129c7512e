"""Dev tool (GPU box): conv1_84_u8_mfma_kernel against the number of observations, nn.Conv2d-layout weights against
operand-order weights.  Usage: python tools/conv84_scaling.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parl_amd import _native as N  # noqa: E402
from parl_amd import ops  # noqa: E402
from conv12_scaling import ev  # noqa: E402

if __name__ == '__main__':
    dev = torch.device('cuda:0')
    w1, b1 = torch.randn(32, 4, 8, 8, device=dev) * 0.1, torch.randn(32, device=dev) * 0.1
    for n in (256, 512, 1024, 2048, 5120):
        obs = torch.randint(0, 256, (n, 4, 84, 84), dtype=torch.uint8, device=dev)
        out = torch.empty((n, 32, 20, 20), dtype=torch.float32, device=dev)
        a = ev(lambda: N.lib().parlhip_atari84_conv1_u8_f32(N.ptr(obs), N.ptr(w1), N.ptr(b1), N.ptr(out), n, N.stream_ptr()))
        b = ev(lambda: ops.atari84_conv1(obs, w1, b1, out=out))
        print('n_obs %5d: %7.1f us, operand-order weights %7.1f us' % (n, a, b))
