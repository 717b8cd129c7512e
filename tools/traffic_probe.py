"""Dev tool (GPU box): launch the scan kernels once each at known algorithmic byte counts, for the
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/prof_traffic.sh.  Also launches calibration
copies with the same access widths (4 B/lane and 16 B/lane) so the gfx950 FETCH_SIZE under-count
(guides/MI355X_MICROARCH.md §HBM) is measured in our own access pattern."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parl_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
out = {}


def flush_cache():
    # 512 MB write: evicts the 256 MB Infinity Cache so that reads below come from HBM
    torch.empty(128 << 20, dtype=torch.float32, device=dev).fill_(1.0)
    torch.cuda.synchronize()


# --- calibration: discounted cumsum with T=1 is a pure copy out[i] = x[i] (1 read + 1 write / elt)
n = 1 << 28  # 1 GiB in, 1 GiB out
x = torch.randn(n + 4, device=dev)
flush_cache()
ops.discount_cumsum(x[:n].view(1, n), 0.9)            # 16 B/lane path (aligned, B % 4 == 0)
out['calib_copy_vec4'] = {'kernel': 'discount_cumsum_kernel<4, 4>', 'read': n * 4, 'write': n * 4}
flush_cache()
ops.discount_cumsum(x[1:n + 1].view(1, n), 0.9)       # 4 B/lane path (base not 16-B aligned)
out['calib_copy_vec1'] = {'kernel': 'discount_cumsum_kernel<1, 8>', 'read': n * 4, 'write': n * 4}
del x
torch.cuda.synchronize()

# --- V-trace from log-probs at the saturating shape: 4 B/lane kernel
T, B = 127, 262144
a = [torch.randn((T, B), device=dev) for _ in range(5)]
boot = torch.randn(B, device=dev)
flush_cache()
ops.vtrace(a[0], a[1], a[2], a[3], a[4], boot)
out['vtrace_T127_B262144'] = {'kernel': 'vtrace_tm_kernel<1, 8', 'read': T * B * 20 + 4 * B, 'write': T * B * 8}
del a
# --- and at B = 1,048,576 (16 B/lane kernel)
T, B = 127, 1 << 20
a = [torch.randn((T, B), device=dev) for _ in range(5)]
boot = torch.randn(B, device=dev)
flush_cache()
ops.vtrace(a[0], a[1], a[2], a[3], a[4], boot)
out['vtrace_T127_B1048576'] = {'kernel': 'vtrace_tm_kernel<4, 4', 'read': T * B * 20 + 4 * B, 'write': T * B * 8}
del a
# --- GAE, PPO storage shape (single-pass look-back plan)
T, B = 2048, 4096
rew, val = torch.randn((T, B), device=dev), torch.randn((T, B), device=dev)
d = (torch.rand((T, B), device=dev) < 0.001).float()
flush_cache()
ops.gae(rew, val, d, torch.randn(B, device=dev), 0.99, 0.95, last_done=torch.zeros(B, device=dev), done_convention=1)
out['gae_T2048_B4096_f32'] = {'kernel': 'gae_lookback_kernel', 'read': T * B * 12, 'write': T * B * 8}
# --- fused V-trace from logits at the BENCH WORKLOAD shape (BASELINE configs[2]: T=50, B=1024, A=6)
T, B, A = 50, 1024, 6
bl, tl = torch.randn((T, B, A), device=dev), torch.randn((T, B, A), device=dev)
act = torch.randint(0, A, (T, B), device=dev)
rw, dn, vl = torch.randn((T, B), device=dev), torch.rand((T, B), device=dev) < 0.01, torch.randn((T, B), device=dev)
flush_cache()
ops.vtrace_from_logits(bl, tl, act, rw, dn, vl, 0.99)
out['vtrace_logits_T50_B1024_A6'] = {'kernel': 'vtrace_logits_', 'read': T * B * (2 * A * 4 + 8 + 4 + 1 + 4),
                                     'write': (T - 1) * B * 8}
# --- the learner's one-kernel loss at the same shape (adds the gradient writes)
flush_cache()
ops.impala_loss(bl, tl, act, rw, dn, vl, 0.99)
out['impala_loss_T50_B1024_A6'] = {'kernel': 'impala_loss_wave_kernel', 'read': T * B * (2 * A * 4 + 8 + 4 + 1 + 4),
                                   'write': (T - 1) * B * 8 + T * B * (4 * A + 4)}
torch.cuda.synchronize()
print(json.dumps(out))
