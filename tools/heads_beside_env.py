"""Dev tool (GPU box): impala_heads_loss at the workload shape (T=50, B=1024, A=6) timed on its own stream WHILE
the emulator kernel runs on another (one wave per SIMD, as in the bench), and alone."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parl_amd import ops  # noqa: E402
from parl_amd.env import DeviceVectorEnv  # noqa: E402

dev = torch.device('cuda')
T, B, A = 50, 1024, 6
hd = torch.relu(torch.randn(T, B, 256, device=dev))
wp, bp = torch.randn(A, 256, device=dev) * 0.1, torch.zeros(A, device=dev)
wv, bv = torch.randn(1, 256, device=dev) * 0.05, torch.zeros(1, device=dev)
bl = torch.randn(T, B, A, device=dev)
ac = torch.randint(0, A, (T, B), device=dev)
rw = torch.randn(T, B, device=dev)
dn = torch.rand(T, B, device=dev) < 0.01
by = T * B * (2048 + A * 4 + 13) + (T - 1) * B * 8
env = DeviceVectorEnv('PongNoFrameskip-v4', 1024, dim=42, horizon=64, seed=1, device=dev)
env.reset()
act = torch.zeros(1024, dtype=torch.int64, device=dev)
rew = torch.zeros(1024, device=dev)
don = torch.zeros(1024, dtype=torch.uint8, device=dev)
sa, sb = torch.cuda.Stream(priority=-1), torch.cuda.Stream()
torch.cuda.synchronize()


def run(with_env, n=120):
    if with_env:
        with torch.cuda.stream(sa):
            for i in range(60):
                if i % 50 == 49:
                    env.roll()
                env.step_async(act, rew, don)
    with torch.cuda.stream(sb):
        for _ in range(5):
            ops.impala_heads_loss(hd, wp, bp, wv, bv, bl, ac, rw, dn, 0.99)
        evs = []
        for _ in range(n):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            ops.impala_heads_loss(hd, wp, bp, wv, bv, bl, ac, rw, dn, 0.99)
            e.record()
            evs.append((s, e))
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    med = ts[len(ts) // 2] * 1e-3
    print('%-22s median %.1f us (%.2f TB/s, frac %.3f), p10 %.1f p90 %.1f' %
          ('beside the env kernel:' if with_env else 'alone:', med * 1e6, by / med / 1e12, by / med / 8e12,
           ts[len(ts) // 10] * 1e3, ts[len(ts) * 9 // 10] * 1e3))


run(False)
run(True)
run(True)
