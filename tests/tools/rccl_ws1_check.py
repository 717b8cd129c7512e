"""Run by tests/test_gpu_dist.py in a subprocess: a ONE-rank RCCL process group (backend "nccl" on
ROCm) on the one GPU of the test box, exercising every collective the data-parallel path issues
(parl_amd/dist.py) on the streams the pipeline issues them from, then a short AsyncActorLearner
run with the gradient all-reduce + small-tensor all-gather on the learner stream, compared with the
same run without a process group (one rank: SUM all-reduce and gather must be the identity)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import parl_amd as parl  # noqa: E402
from parl_amd import dist as pdist  # noqa: E402
from parl_amd.env import DeviceVectorEnv  # noqa: E402
from parl_amd.models import AtariModel42  # noqa: E402
from parl_amd.rollout import AsyncActorLearner  # noqa: E402


def run(use_dist, steps=3):
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    E, T = 16, 8
    env = DeviceVectorEnv('PongNoFrameskip-v4', E, dim=42, horizon=T, seed=3, device=dev)
    model = AtariModel42(env.act_dim).to(dev)
    alg = parl.algorithms.IMPALA(model, sample_batch_steps=T, gamma=0.99, vf_loss_coeff=0.5,
                                 clip_rho_threshold=1.0, clip_pg_rho_threshold=1.0)
    if use_dist:
        pdist.broadcast_model(model)
        alg.grad_hook = pdist.FlatGradAllReduce(model)
    pipe = AsyncActorLearner(alg, [env], T, seed=9)
    pipe.gather_small = use_dist
    pipe.prime()
    losses = []
    for _ in range(steps):
        loss, kl = pipe.step(1e-3, -0.01)
        pipe.wait_outputs()
        losses.append(float(loss.total_loss.item()))
        if use_dist:
            g = pipe.gathered[0]
            assert g['rewards'].shape == (1, T * E) and g['actions'].dtype == torch.int64
    pipe.synchronize()
    return losses, [p.detach().cpu().numpy().copy() for p in model.parameters()]


def main():
    assert torch.cuda.is_available()
    ref_losses, ref_w = run(False)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', sys.argv[1] if len(sys.argv) > 1 else '29544')
    rank, local, world = pdist.init(backend='nccl', force=True)
    assert pdist.active() and world == 1 and torch.distributed.get_backend() == 'nccl'
    dev = torch.device('cuda', 0)
    # every collective on a side stream (actor-priority and learner-priority), checked for identity
    for prio in (-1, 0):
        st = torch.cuda.Stream(device=dev, priority=prio)
        with torch.cuda.stream(st):
            m = torch.nn.Linear(64, 32).to(dev)
            hook = pdist.FlatGradAllReduce(m)
            hook.zero_grad()
            m(torch.randn(8, 64, device=dev)).square().sum().backward()
            before = hook.flat.clone()
            assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(hook.params, hook.views))
            hook(m)
            assert torch.equal(hook.flat, before)
            avg = pdist.FlatGradAllReduce(m, average=True)
            avg.zero_grad()
            m(torch.randn(8, 64, device=dev)).sum().backward()
            b2 = avg.flat.clone()
            avg(m)
            assert torch.equal(avg.flat, b2)
            x = {'a': torch.arange(12, device=dev).reshape(3, 4), 'b': torch.rand(5, device=dev),
                 'c': (torch.rand(7, device=dev) > 0.5).to(torch.uint8)}
            g = pdist.all_gather_small(x)
            assert all(torch.equal(g[k][0], x[k]) and g[k].shape[0] == 1 for k in x)
            # two env groups gather tensors of identical keys and shapes in one update: each slot owns
            # its receive buffers, the first result must survive the second call
            y = {k: v + 1 for k, v in x.items()}
            g0, g1 = pdist.all_gather_small(x, slot=0), pdist.all_gather_small(y, slot=1)
            assert all(g0[k].data_ptr() != g1[k].data_ptr() for k in x)
            assert all(torch.equal(g0[k][0], x[k]) and torch.equal(g1[k][0], y[k]) for k in x)
            pdist.broadcast_model(m)
        st.synchronize()
    pdist.barrier()
    assert abs(pdist.all_reduce_max_scalar(1.25) - 1.25) < 1e-12
    losses, w = run(True)
    np.testing.assert_allclose(losses, ref_losses, rtol=1e-6)
    for a, b in zip(w, ref_w):
        np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-7)
    torch.distributed.destroy_process_group()
    print('RCCL_WS1_OK', losses)


if __name__ == '__main__':
    main()
