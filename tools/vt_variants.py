"""Dev tool (GPU box): time V-trace launch variants (PARLHIP_EXP_VT) at the saturating shapes."""
import os, subprocess, sys
code = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
from parl_amd import ops
dev = torch.device('cuda')
for T, B in ((127, 262144), (127, 1048576)):
    x = [torch.randn((T, B), device=dev) for _ in range(5)]
    boot = torch.randn(B, device=dev)
    for _ in range(3): ops.vtrace(x[0], x[1], x[2], x[3], x[4], boot)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): ops.vtrace(x[0], x[1], x[2], x[3], x[4], boot)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    by = T * B * 28 + 4 * B
    print('VT=%s T=%d B=%d: %.1f us  %.0f GB/s' % (os.environ.get('PARLHIP_EXP_VT', 'default'), T, B, ms * 1e3, by / ms / 1e6))
    del x
'''
for v in [None] + list(range(0, 9)):
    env = dict(os.environ)
    if v is not None:
        env['PARLHIP_EXP_VT'] = str(v)
    subprocess.run([sys.executable, '-c', code], env=env)
