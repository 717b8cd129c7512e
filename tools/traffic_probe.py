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
del bl, tl, act, rw, dn, vl
# --- frame_post at the bench shape (1024 envs, 42x42): two 33,600 B colour frames in, 1,764 B out per env
from parl_amd.env import DeviceVectorEnv  # noqa: E402
env = DeviceVectorEnv('PongNoFrameskip-v4', 1024, dim=42, horizon=8, seed=1, device=dev)
env.reset()
env.step_async(torch.zeros(1024, dtype=torch.int64, device=dev))
flush_cache()
env._frame_post(2)
out['frame_post_E1024_d42'] = {'kernel': 'frame_post_kernel', 'read': 1024 * 2 * 33600, 'write': 1024 * (1764 + 1)}
del env
# --- the conv kernels (compute-bound; traffic = are activations / observations read once?)
n = 8192
obs = torch.randint(0, 256, (n, 4, 42, 42), dtype=torch.uint8, device=dev)
w1, b1 = torch.randn(16, 4, 4, 4, device=dev) * 0.2, torch.zeros(16, device=dev)
w2, b2 = torch.randn(32, 16, 4, 4, device=dev) * 0.1, torch.zeros(32, device=dev)
flush_cache()
a2 = ops.atari42_conv12(obs, w1, b1, w2, b2)
out['conv12_fwd_n8192'] = {'kernel': 'conv12_u8_mfma_kernel<false, false, false>', 'read': n * 7056, 'write': n * 15488}
dy = torch.randn((n, 3872), device=dev)
flush_cache()
ops.atari42_conv12_backward(obs, w1, b1, w2, a2, dy)
out['conv12_bwd_n8192'] = {'kernel': 'conv12_bwd_u8_mfma_kernel<false, false>', 'read': n * (7056 + 2 * 15488), 'write': 512 * 9264 * 4}
# the learner's pair of round 6: the forward also writes the padded conv1 tile (40,000 B per observation), the backward
# reads it back (matched on the template arguments: <false, true, true> / <true, true>)
pk = ops.atari42_conv12_pack(w1, w2)
flush_cache()
a2, a1s = ops.atari42_conv12(obs, w1, b1, w2, b2, packed=pk, save_a1=True)
out['conv12_fwd_save_n8192'] = {'kernel': 'conv12_u8_mfma_kernel<false, true, true>', 'read': n * 7056, 'write': n * (15488 + 40000)}
flush_cache()
ops.atari42_conv12_backward(obs, w1, b1, w2, a2, dy, packed=pk, a1=a1s)
out['conv12_bwd_saved_n8192'] = {'kernel': 'conv12_bwd_u8_mfma_kernel<true, true>', 'read': n * (7056 + 2 * 15488 + 39936),
                                 'write': 512 * 9264 * 4}
del obs, a2, dy, a1s
a1 = torch.relu(torch.randn(n, 32, 20, 20, device=dev))
w2c, w3c = torch.randn(64, 32, 4, 4, device=dev) * 0.05, torch.randn(64, 64, 3, 3, device=dev) * 0.05
z = torch.zeros(64, device=dev)
flush_cache()
ops.atari84_conv23(a1, w2c, z, w3c, z)
out['conv23_84_fwd_n8192'] = {'kernel': 'conv23_84_mfma_kernel', 'read': n * 51200, 'write': n * 20736}
# --- heads + loss + heads' backward at the workload shape
T, B, A = 50, 1024, 6
hd = torch.relu(torch.randn(T, B, 256, device=dev))
wp, bp = torch.randn(A, 256, device=dev) * 0.1, torch.zeros(A, device=dev)
wv, bv = torch.randn(1, 256, device=dev) * 0.05, torch.zeros(1, device=dev)
bl = torch.randn(T, B, A, device=dev)
ac = torch.randint(0, A, (T, B), device=dev)
rw = torch.randn(T, B, device=dev)
dn = torch.rand(T, B, device=dev) < 0.01
flush_cache()
ops.impala_heads_loss(hd, wp, bp, wv, bv, bl, ac, rw, dn, 0.99)
out['impala_heads_loss_T50_B1024_A6'] = {'kernel': 'impala_heads_loss_q_kernel',
                                         'read': T * B * (1024 + A * 4 + 8 + 4 + 1), 'write': T * B * 1024 + (T - 1) * B * 8}
torch.cuda.synchronize()
print(json.dumps(out))
