"""Dev tool (GPU box): engine clock (rocm-smi) while ONE kernel loops — does the clock the latency-bound emulator runs at
depend on what runs beside it?  Usage: python tools/clock_probe.py"""
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parl_amd import ops  # noqa: E402
from parl_amd.env import DeviceVectorEnv  # noqa: E402

dev = torch.device('cuda')
R = 1000
obs = torch.randint(0, 256, (R, 4, 42, 42), dtype=torch.uint8, device=dev)
w1, b1 = torch.randn(16, 4, 4, 4, device=dev) * 0.2, torch.zeros(16, device=dev)
w2, b2 = torch.randn(32, 16, 4, 4, device=dev) * 0.1, torch.zeros(32, device=dev)
pk = ops.atari42_conv12_pack(w1, w2)
a2 = ops.atari42_conv12(obs, w1, b1, w2, b2, packed=pk)
dy = torch.randn_like(a2)
W3 = torch.randn(256, 3872, device=dev) * 0.01
env = DeviceVectorEnv('PongNoFrameskip-v4', 1024, dim=42, horizon=64, seed=1, device=dev)
env.reset()
act = torch.zeros(1024, dtype=torch.int64, device=dev)
rew, don = torch.zeros(1024, device=dev), torch.zeros(1024, dtype=torch.uint8, device=dev)


def env_steps():
    if env.t >= env.horizon:
        env.roll()
    env.step_async(act, rew, don)


def sample(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True, timeout=5).stdout
            m = re.search(r'sclk clock level: \S+ \((\d+)Mhz\)', txt)
            p = re.search(r'Power \(W\): ([\d.]+)', txt)
            out.append((int(m.group(1)) if m else None, float(p.group(1)) if p else None))
        except Exception as e:  # noqa
            out.append((None, None))
        time.sleep(0.05)


for name, fn in (('idle', None), ('env steps', env_steps),
                 ('conv12 forward 1000 rows', lambda: ops.atari42_conv12(obs, w1, b1, w2, b2, packed=pk)),
                 ('conv12 backward 1000 rows', lambda: ops.atari42_conv12_backward(obs, w1, b1, w2, a2, dy, packed=pk)),
                 ('GEMM [1000,3872]x[3872,256]', lambda: torch.mm(a2, W3.t()))):
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out))
    th.start()
    t0 = time.time()
    with torch.no_grad():
        while time.time() - t0 < 2.5:
            if fn is None:
                time.sleep(0.05)
            else:
                for _ in range(50):
                    fn()
                torch.cuda.synchronize()
    stop.set()
    th.join()
    clk = [c for c, _ in out if c]
    pw = [p for _, p in out if p]
    print('%-30s sclk MHz %s  power W %s  (%d samples)' % (name, (min(clk), sorted(clk)[len(clk) // 2], max(clk)) if clk else None,
                                                          (min(pw), sorted(pw)[len(pw) // 2], max(pw)) if pw else None, len(out)))
