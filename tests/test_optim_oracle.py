"""oracle/optim_oracle.py (the CPU restatement parlhip_clip_adam_f32 is checked against on the GPU,
tests/test_gpu_optim.py) pinned on the functions the reference itself calls: torch.nn.utils.clip_grad_norm_ +
torch.optim.Adam on the host (parl/algorithms/torch/a2c.py:76-78)."""
import numpy as np
import pytest
import torch

from oracle import optim_oracle

SHAPES = [(16, 4, 4, 4), (16, ), (32, 16, 4, 4), (32, ), (64, 300), (6, 256), (6, ), (1, 256), (1, ), (2049, ), (1, )]


@pytest.mark.parametrize('grad_scale', [0.01, 30.0])   # clip inactive / active at max_norm 40
def test_oracle_matches_torch_clip_grad_norm_and_adam(grad_scale):
    g = torch.Generator().manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(s, generator=g) * 0.1) for s in SHAPES]
    opt = torch.optim.Adam(ps, lr=1e-3)
    P = [p.detach().numpy().copy() for p in ps]
    M = [np.zeros_like(x) for x in P]
    V = [np.zeros_like(x) for x in P]
    S = [0.0] * len(P)
    for it in range(6):
        lr = 1e-3 if it < 3 else 5e-4
        for grp in opt.param_groups:
            grp['lr'] = lr
        grads = [torch.randn(s, generator=g) * grad_scale for s in SHAPES]
        G = [x.numpy().copy() for x in grads]
        for p, x in zip(ps, grads):
            p.grad = x.clone()
        norm_t = float(torch.nn.utils.clip_grad_norm_(ps, max_norm=40.0))
        opt.step()
        norm_o = optim_oracle.clip_adam_step(P, G, M, V, S, lr, max_norm=40.0)
        assert abs(norm_o - norm_t) <= 2e-6 * norm_t
        assert (norm_t > 40.0) == (grad_scale > 1.0)
        for i, p in enumerate(ps):
            st = opt.state[p]
            assert float(st['step']) == S[i] == it + 1
            np.testing.assert_allclose(G[i], p.grad.numpy(), rtol=3e-6, atol=0)
            np.testing.assert_allclose(M[i], st['exp_avg'].numpy(), rtol=1e-5, atol=1e-6 * grad_scale)
            np.testing.assert_allclose(V[i], st['exp_avg_sq'].numpy(), rtol=1e-5, atol=1e-7 * grad_scale ** 2)
            np.testing.assert_allclose(P[i], p.detach().numpy(), rtol=0, atol=2e-7)
