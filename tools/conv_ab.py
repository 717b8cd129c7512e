"""Dev tool (GPU box): every hand-written conv kernel of the two Atari networks, HIP-event timed (median of 30) at the
shapes the pipelines run them at — for A/B runs of two builds of libparl_hip.so in one call (PARL_HIP_LIB selects
the build).  Usage: [PARL_HIP_LIB=build_exp/x.so] python tools/conv_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parl_amd import ops  # noqa: E402

F32_MFMA_PEAK = 157.3e12


def ev(fn, iters=30):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in e:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in e)
    return t[len(t) // 2] * 1e3


def row(name, us, flops=None):
    extra = ''
    if flops:
        extra = '  %.1f TFLOP/s = %.2f of the f32 MFMA peak' % (flops / us / 1e6, flops / (us * 1e-6) / F32_MFMA_PEAK)
    print('%-44s %9.1f us%s' % (name, us, extra), flush=True)


def main():
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    print('library:', os.environ.get('PARL_HIP_LIB', 'in-tree libparl_hip.so'))
    w1, b1 = torch.randn(16, 4, 4, 4, device=dev) * 0.2, torch.randn(16, device=dev) * 0.1
    w2, b2 = torch.randn(32, 16, 4, 4, device=dev) * 0.1, torch.randn(32, device=dev) * 0.1
    pk = ops.atari42_conv12_pack(w1, w2)
    f12 = 2.0 * (441 * 16 * 64 + 121 * 32 * 256)
    b12 = 2.0 * 16 * 2812 * 64          # issued MFMA work of the backward kernel per observation (16*16*4 FMAs each)
    for n in (1000, 1024, 8000, 51200):
        obs = torch.randint(0, 256, (n, 4, 42, 42), dtype=torch.uint8, device=dev)
        out = torch.empty((n, 3872), device=dev)
        row('conv12 forward, %d rows' % n, ev(lambda: ops.atari42_conv12(obs, w1, b1, w2, b2, out=out, packed=pk)), n * f12)
        if n != 1024:
            a2 = ops.atari42_conv12(obs, w1, b1, w2, b2, packed=pk)
            dy = torch.randn_like(a2)
            row('conv12 backward (recompute), %d rows' % n,
                ev(lambda: ops.atari42_conv12_backward(obs, w1, b1, w2, a2, dy, packed=pk)), n * b12)
            if n <= 8192 and hasattr(ops, 'A1_SAVE_MAX_ROWS'):
                a1 = torch.empty((n, 10000), device=dev)
                row('conv12 forward + a1 saved, %d rows' % n,
                    ev(lambda: ops.atari42_conv12(obs, w1, b1, w2, b2, out=out, packed=pk, save_a1=True)), n * f12)
                _, a1 = ops.atari42_conv12(obs, w1, b1, w2, b2, packed=pk, save_a1=True)
                row('conv12 backward (a1 read back), %d rows' % n,
                    ev(lambda: ops.atari42_conv12_backward(obs, w1, b1, w2, a2, dy, packed=pk, a1=a1)),
                    n * 2.0 * 16 * (2812 - 448) * 64)
        del obs, out
    # ---- the 84x84 network (examples/A2C/atari_model.py) ----
    c1w, c1b = torch.randn(32, 4, 8, 8, device=dev) * 0.05, torch.randn(32, device=dev) * 0.1
    c2w, c2b = torch.randn(64, 32, 4, 4, device=dev) * 0.05, torch.randn(64, device=dev) * 0.1
    c3w, c3b = torch.randn(64, 64, 3, 3, device=dev) * 0.05, torch.randn(64, device=dev) * 0.1
    wt1 = ops.atari84_conv1_layout(c1w)
    wt23 = ops.atari84_conv23_layouts(c2w, c3w)
    for n in (256, 1024, 5120):
        obs = torch.randint(0, 256, (n, 4, 84, 84), dtype=torch.uint8, device=dev)
        a1 = torch.empty((n, 32, 20, 20), device=dev)
        with torch.no_grad():
            row('conv1_84 forward, %d rows' % n, ev(lambda: ops.atari84_conv1(obs, c1w, c1b, out=a1, wt1=wt1)),
                n * 2.0 * 400 * 32 * 256)
            row('conv23_84 forward, %d rows' % n,
                ev(lambda: ops.atari84_conv23(a1, c2w, c2b, c3w, c3b, save_a2=True, wt23=wt23)),
                n * 2.0 * (121 * 64 * 512 + 81 * 64 * 576))
            if n == 5120:
                a3, a2 = ops.atari84_conv23(a1, c2w, c2b, c3w, c3b, save_a2=True, wt23=wt23)
                g3 = torch.randn_like(a3)
                row('conv3_84 backward, %d rows' % n, ev(lambda: ops.atari84_conv3_backward(a2, a3, g3, c3w)),
                    n * 2.0 * 2 * 81 * 64 * 576)
                dz2 = ops.atari84_conv3_backward(a2, a3, g3, c3w)[0]
                row('conv2_84 backward, %d rows' % n, ev(lambda: ops.atari84_conv2_backward(a1, dz2, c2w)),
                    n * 2.0 * 2 * 121 * 64 * 512)
                dz1 = ops.atari84_conv2_backward(a1, dz2, c2w)[0]
                row('conv1_84 backward, %d rows' % n, ev(lambda: ops.atari84_conv1_backward(obs, dz1)),
                    n * 2.0 * 400 * 32 * 256)
        del obs, a1


if __name__ == '__main__':
    main()
