"""calc_gae / calc_discount_sum_rewards with the reference's signatures
(parl/utils/rl_utils.py:21-51), computed by the gfx950 scan kernels.

The reference functions take one (env, segment) at a time as Python lists on the host; these
drop-ins keep that contract (host in, float64 numpy out) by staging the segment as a [T,1] batch
on the GPU — correct but latency-bound.  The batched device API is parl_amd.ops.gae /
ops.discount_cumsum ([T,B] tensors), which is what the on-device A2C/PPO paths use.
Precision: the kernels compute in float32 and the result is widened to float64; the reference is
float64 end to end (rl_utils.py:49-50).  The two agree to 1e-5 (tests/test_gpu_scans.py, fixtures
produced by the reference function) — nothing hot calls these shims.
There is no CPU fallback: without the HIP library or a GPU these raise."""
import numpy as np
import torch

from .. import ops

__all__ = ['calc_discount_sum_rewards', 'calc_gae']


def _dev():
    if not torch.cuda.is_available():
        raise RuntimeError('parl_amd.utils.rl_utils needs an MI355X (no CPU fallback)')
    return torch.device('cuda')


def calc_discount_sum_rewards(rewards, gamma):
    x = torch.as_tensor(np.asarray(rewards, dtype=np.float32).reshape(-1, 1), device=_dev())
    return ops.discount_cumsum(x, gamma).reshape(-1).double().cpu().numpy()


def calc_gae(rewards, values, next_value, gamma, lam):
    dev = _dev()
    r = torch.as_tensor(np.asarray(rewards, dtype=np.float32).reshape(-1, 1), device=dev)
    v = torch.as_tensor(np.asarray(values, dtype=np.float32).reshape(-1, 1), device=dev)
    nv = torch.as_tensor(np.asarray(next_value, dtype=np.float32).reshape(-1)[:1], device=dev)
    d = torch.zeros(r.shape, dtype=torch.uint8, device=dev)
    adv, _ = ops.gae(r, v, d, nv, gamma, lam)
    return adv.reshape(-1).double().cpu().numpy()
