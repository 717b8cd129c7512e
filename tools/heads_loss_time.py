import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parl_amd import ops
dev = torch.device('cuda')
T, B, A = 50, 1024, 6
hd = torch.relu(torch.randn(T, B, 256, device=dev))
wp, bp = torch.randn(A, 256, device=dev) * 0.1, torch.zeros(A, device=dev)
wv, bv = torch.randn(1, 256, device=dev) * 0.05, torch.zeros(1, device=dev)
bl = torch.randn(T, B, A, device=dev); ac = torch.randint(0, A, (T, B), device=dev)
rw = torch.randn(T, B, device=dev); dn = torch.rand(T, B, device=dev) < 0.01
for _ in range(5): ops.impala_heads_loss(hd, wp, bp, wv, bv, bl, ac, rw, dn, 0.99)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(50): ops.impala_heads_loss(hd, wp, bp, wv, bv, bl, ac, rw, dn, 0.99)
e.record(); torch.cuda.synchronize()
by = T * B * (2048 + A * 4 + 13) + (T - 1) * B * 8
t = s.elapsed_time(e) / 50 * 1e-3
print('heads_loss standalone (incl. python + partial sum): %.1f us, %.2f TB/s algorithmic' % (t * 1e6, by / t / 1e12))
