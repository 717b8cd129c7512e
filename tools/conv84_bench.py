"""Dev tool (GPU box): the actors' conv1 of the 84x84 model (parlhip_atari84_conv1_u8_f32), HIP-event timed.
PARL_HIP_LIB selects the library (A/B of kernel variants)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parl_amd import ops  # noqa: E402
from tools.microbench import timeit  # noqa: E402

if __name__ == '__main__':
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    w, b = torch.randn(32, 4, 8, 8, device=dev) * 0.05, torch.randn(32, device=dev) * 0.1
    for n in (256, 1024, 5120):
        obs = torch.randint(0, 256, (n, 4, 84, 84), dtype=torch.uint8, device=dev)
        out = torch.empty((n, 32, 20, 20), device=dev)
        s = timeit(lambda: ops.atari84_conv1(obs, w, b, out=out), iters=50)
        ref = torch.relu(torch.nn.functional.conv2d(obs[:8].double() / 255.0, w.double(), b.double(), stride=4, padding=1))
        err = float((out[:8].double() - ref).abs().max())
        print('%s conv1_84 n=%d: %.1f us  %.2f TFLOP/s  max err vs fp64 %.2e' %
              (os.path.basename(os.environ.get('PARL_HIP_LIB', 'libparl_hip.so')), n, s * 1e6, n * 6.5536e6 / s / 1e12, err))
