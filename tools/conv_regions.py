"""Dev tool (GPU box): where conv12_bwd_u8_mfma_kernel's time goes, per phase — diagnostic build
(tools/build_conv_variant.sh convreg -DPARLHIP_CONV_REGIONS; PARL_HIP_LIB=build_exp/convreg.so): s_memtime clocks of
wave 0 of every workgroup per phase, per observation."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parl_amd import _native as N  # noqa: E402
from parl_amd import ops  # noqa: E402

if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 51200
    dev = torch.device('cuda:0')
    f = N.lib().parlhip_debug_conv_regions
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_int]
    obs = torch.randint(0, 256, (n, 4, 42, 42), dtype=torch.uint8, device=dev)
    w1, b1 = torch.randn(16, 4, 4, 4, device=dev) * 0.2, torch.randn(16, device=dev) * 0.1
    w2, b2 = torch.randn(32, 16, 4, 4, device=dev) * 0.1, torch.randn(32, device=dev) * 0.1
    out = torch.empty((n, 3872), device=dev)
    saved = len(sys.argv) > 2 and sys.argv[2] == 'a1'   # the round-6 pair: conv1 saved by the forward, read back here
    pk = ops.atari42_conv12_pack(w1, w2)
    a1 = None
    if saved:
        _, a1 = ops.atari42_conv12(obs, w1, b1, w2, b2, out=out, packed=pk, save_a1=True)
    else:
        ops.atari42_conv12(obs, w1, b1, w2, b2, out=out, packed=pk)
    dy = torch.randn((n, 3872), device=dev)
    buf = np.zeros(16, np.uint64)
    for it in range(3):
        torch.cuda.synchronize()
        f(buf.ctypes.data, 1)
        ops.atari42_conv12_backward(obs, w1, b1, w2, out, dy, packed=pk, a1=a1)
        torch.cuda.synchronize()
        f(buf.ctypes.data, 0)
    names = ['barrier at the top', 'fill', '(1) conv1 recompute' if not saved else '(1) -', '(2) dW2', '(3) dz1', '(4) dW1']
    print('a1 read back' if saved else 'conv1 recomputed')
    nobs = float(buf[15])
    tot = float(buf[:6].sum())
    print('conv12_bwd n=%d: %.0f clocks per observation (wave 0 of each workgroup)' % (n, tot / nobs))
    for i, nm in enumerate(names):
        print('  %-22s %8.0f clocks  %5.1f %%' % (nm, buf[i] / nobs, 100.0 * buf[i] / tot))
