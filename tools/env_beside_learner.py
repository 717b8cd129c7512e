"""Dev tool (GPU box): what ONE kind of learner kernel costs the emulator.  The env launch (VectorEnv.step with its
observation, E = 1024, 42x42) is event-timed on a high-priority stream while another stream loops over one learner
kernel at the shape of a 1000-row update; alone first.  Usage: python tools/env_beside_learner.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parl_amd import ops  # noqa: E402
from parl_amd.env import DeviceVectorEnv  # noqa: E402

dev = torch.device('cuda')
E, R = 1024, 1000
env = DeviceVectorEnv('PongNoFrameskip-v4', E, dim=42, horizon=64, seed=1, device=dev)
env.reset()
act = torch.zeros(E, dtype=torch.int64, device=dev)
rew, don = torch.zeros(E, device=dev), torch.zeros(E, dtype=torch.uint8, device=dev)
for _ in range(30):
    env.step_async(act, rew, don)
env.roll()
sa, sb = torch.cuda.Stream(priority=-1), torch.cuda.Stream()

torch.manual_seed(0)
obs = torch.randint(0, 256, (R, 4, 42, 42), dtype=torch.uint8, device=dev)
w1, b1 = torch.randn(16, 4, 4, 4, device=dev) * 0.2, torch.zeros(16, device=dev)
w2, b2 = torch.randn(32, 16, 4, 4, device=dev) * 0.1, torch.zeros(32, device=dev)
pk = ops.atari42_conv12_pack(w1, w2)
a2 = ops.atari42_conv12(obs, w1, b1, w2, b2, packed=pk)
dy = torch.randn_like(a2)
W3 = torch.randn(256, 3872, device=dev) * 0.01
h = torch.relu(torch.randn(R, 256, device=dev))
dh = torch.randn(R, 256, device=dev)
big_a, big_b = torch.randn(64 << 20, device=dev), torch.empty(64 << 20, device=dev)
T, B, A = 50, 20, 6
hd = torch.relu(torch.randn(T, B, 256, device=dev))
wp, bp = torch.randn(A, 256, device=dev) * 0.1, torch.zeros(A, device=dev)
wv, bv = torch.randn(1, 256, device=dev) * 0.05, torch.zeros(1, device=dev)
bl, ac = torch.randn(T, B, A, device=dev), torch.randint(0, A, (T, B), device=dev)
rw, dn = torch.randn(T, B, device=dev), torch.rand(T, B, device=dev) < 0.01

KERNELS = {
    'nothing': None,
    'conv12 forward, 1000 rows': lambda: ops.atari42_conv12(obs, w1, b1, w2, b2, packed=pk),
    'conv12 backward, 1000 rows': lambda: ops.atari42_conv12_backward(obs, w1, b1, w2, a2, dy, packed=pk),
    'trunk GEMM forward [1000,3872]x[3872,256]': lambda: torch.mm(a2, W3.t()),
    'trunk GEMM dW [256,1000]x[1000,3872]': lambda: torch.mm(dh.t(), a2),
    'trunk GEMM dX [1000,256]x[256,3872]': lambda: torch.mm(dh, W3),
    'heads loss, 20 sequences': lambda: ops.impala_heads_loss(hd, wp, bp, wv, bv, bl, ac, rw, dn, 0.99),
    'HBM copy 256 MB': lambda: big_b.copy_(big_a),
}


def run(name, fn, steps=40):
    torch.cuda.synchronize()
    if fn is not None:
        with torch.cuda.stream(sb):
            for _ in range(3000 if 'copy' not in name else 400):
                fn()
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
    print('%-46s env launch median %.1f us (p10 %.1f, p90 %.1f)%s' %
          (name + ':', ts[len(ts) // 2] * 1e3, ts[len(ts) // 10] * 1e3, ts[len(ts) * 9 // 10] * 1e3,
           '' if (fn is None or busy) else '   [the other stream ran dry before the env steps ended]'))


only = os.environ.get('ENV_BESIDE_ONLY')
if only:
    KERNELS = {k: v for k, v in KERNELS.items() if k == 'nothing' or only in k}
with torch.no_grad():
    for k, f in KERNELS.items():
        if f is not None:
            for _ in range(3):
                f()
    for k, f in KERNELS.items():
        run(k, f)
