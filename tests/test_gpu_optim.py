"""ops.ClipAdam (parlhip_clip_adam_f32: global-norm clip + Adam in two launches) against the pair it replaces in the
graphed learner — torch.nn.utils.clip_grad_norm_ + torch.optim.Adam(capturable, fused).step(), the host mirror of
parl/algorithms/paddle/impala.py:113-117 (Adam + ClipGradByGlobalNorm(40)) — and against the CPU oracle
(oracle/optim_oracle.py).  -m gpu."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _models(shapes, dev, seed):
    g = torch.Generator(device='cpu').manual_seed(seed)
    ps = [torch.nn.Parameter((torch.randn(s, generator=g) * 0.1).to(dev)) for s in shapes]
    return ps, [torch.nn.Parameter(p.detach().clone()) for p in ps]


@pytest.mark.parametrize('grad_scale', [0.01, 30.0])   # clip inactive / active (max_norm 40)
def test_clip_adam_matches_torch_clip_and_fused_adam(grad_scale):
    from parl_amd import ops
    from parl_amd.algorithms.impala.graphed import make_capturable, set_lr
    dev = torch.device('cuda', 0)
    # the 42x42 Atari model's parameter shapes plus odd sizes (not a multiple of the 2,048-element chunk, one element)
    shapes = [(16, 4, 4, 4), (16, ), (32, 16, 4, 4), (32, ), (256, 3872), (256, ), (6, 256), (6, ), (1, 256), (1, ),
              (2049, ), (4095, 3), (1, )]
    pa, pb = _models(shapes, dev, 0)
    oa, ob = torch.optim.Adam(pa, lr=1e-3), torch.optim.Adam(pb, lr=1e-3)
    make_capturable(oa, dev)
    make_capturable(ob, dev)
    assert ops.ClipAdam.supported(oa)
    ca = ops.ClipAdam(oa, 40.0)
    g = torch.Generator(device='cpu').manual_seed(1)
    for it in range(6):
        lr = 1e-3 if it < 3 else 5e-4
        set_lr(oa, lr)
        set_lr(ob, lr)
        grads = [(torch.randn(s, generator=g) * grad_scale).to(dev) for s in shapes]
        for p, q, gr in zip(pa, pb, grads):
            p.grad, q.grad = gr.clone(), gr.clone()
        ca.step()
        norm = torch.nn.utils.clip_grad_norm_(pb, max_norm=40.0)
        ob.step()
        torch.cuda.synchronize()
        assert abs(float(ca.norm) - float(norm)) <= 2e-6 * float(norm)
        assert (float(norm) > 40.0) == (grad_scale > 1.0)
        for p, q in zip(pa, pb):
            sa, sb = oa.state[p], ob.state[q]
            assert float(sa['step']) == float(sb['step']) == it + 1
            torch.testing.assert_close(p.grad, q.grad, rtol=3e-6, atol=0)           # clipped in place, as torch does
            torch.testing.assert_close(sa['exp_avg'], sb['exp_avg'], rtol=1e-5, atol=1e-6 * grad_scale)   # (float rounding of
            #                                                             m + w (g - m) where the two nearly cancel)
            torch.testing.assert_close(sa['exp_avg_sq'], sb['exp_avg_sq'], rtol=1e-5, atol=1e-7 * grad_scale ** 2)
            torch.testing.assert_close(p.detach(), q.detach(), rtol=0, atol=2e-7)    # one step moves a weight by <= lr
    # deterministic: the same six steps again give the same bits
    pc, _ = _models(shapes, dev, 0)
    oc = torch.optim.Adam(pc, lr=1e-3)
    make_capturable(oc, dev)
    cc = ops.ClipAdam(oc, 40.0)
    g = torch.Generator(device='cpu').manual_seed(1)
    for it in range(6):
        set_lr(oc, 1e-3 if it < 3 else 5e-4)
        for p, s in zip(pc, shapes):
            p.grad = (torch.randn(s, generator=g) * grad_scale).to(dev)
        cc.step()
    torch.cuda.synchronize()
    for p, q in zip(pa, pc):
        assert torch.equal(p.detach(), q.detach())


@pytest.mark.parametrize('grad_scale', [0.01, 30.0])
def test_clip_adam_matches_the_cpu_oracle(grad_scale):
    """the same kernels against oracle/optim_oracle.py (numpy float32 restatement, pinned on torch's host
    clip_grad_norm_ + Adam in tests/test_optim_oracle.py) on the same seeded inputs: six steps, a changing learning
    rate, tensors that are no multiple of the 2,048-element chunk"""
    import numpy as np
    from oracle import optim_oracle
    from parl_amd import ops
    from parl_amd.algorithms.impala.graphed import make_capturable, set_lr
    dev = torch.device('cuda', 0)
    shapes = [(16, 4, 4, 4), (16, ), (32, 16, 4, 4), (32, ), (256, 3872), (6, 256), (6, ), (1, 256), (1, ), (2049, ),
              (4095, 3), (1, )]
    pa, _ = _models(shapes, dev, 0)
    oa = torch.optim.Adam(pa, lr=1e-3)
    make_capturable(oa, dev)
    ca = ops.ClipAdam(oa, 40.0)
    P = [p.detach().cpu().numpy().copy() for p in pa]
    M, V, S = [np.zeros_like(x) for x in P], [np.zeros_like(x) for x in P], [0.0] * len(P)
    g = torch.Generator(device='cpu').manual_seed(1)
    for it in range(6):
        lr = 1e-3 if it < 3 else 5e-4
        set_lr(oa, lr)
        grads = [torch.randn(s, generator=g) * grad_scale for s in shapes]
        G = [x.numpy().copy() for x in grads]
        for p, gr in zip(pa, grads):
            p.grad = gr.to(dev)
        ca.step()
        norm_o = optim_oracle.clip_adam_step(P, G, M, V, S, lr, max_norm=40.0)
        torch.cuda.synchronize()
        assert abs(float(ca.norm) - norm_o) <= 4e-6 * norm_o
        for i, p in enumerate(pa):
            st = oa.state[p]
            assert float(st['step']) == S[i] == it + 1
            np.testing.assert_allclose(p.grad.cpu().numpy(), G[i], rtol=6e-6, atol=0)
            np.testing.assert_allclose(st['exp_avg'].cpu().numpy(), M[i], rtol=2e-5, atol=2e-6 * grad_scale)
            np.testing.assert_allclose(st['exp_avg_sq'].cpu().numpy(), V[i], rtol=2e-5, atol=2e-7 * grad_scale ** 2)
            np.testing.assert_allclose(p.detach().cpu().numpy(), P[i], rtol=0, atol=4e-7)


def test_clip_adam_covers_what_it_says():
    from parl_amd import ops
    from parl_amd.algorithms.impala.graphed import make_capturable
    dev = torch.device('cuda', 0)
    ps = [torch.nn.Parameter(torch.zeros(8, device=dev)) for _ in range(3)]
    o = torch.optim.Adam(ps, lr=1e-3)
    assert not ops.ClipAdam.supported(o)                        # lr is a Python float: not capturable yet
    make_capturable(o, dev)
    assert ops.ClipAdam.supported(o)
    assert not ops.ClipAdam.supported(torch.optim.Adam(ps, lr=torch.tensor(1e-3, device=dev), weight_decay=0.1))
    assert not ops.ClipAdam.supported(torch.optim.AdamW(ps, lr=torch.tensor(1e-3, device=dev)))
    many = [torch.nn.Parameter(torch.zeros(2, device=dev)) for _ in range(17)]
    om = torch.optim.Adam(many, lr=1e-3)
    make_capturable(om, dev)
    assert not ops.ClipAdam.supported(om)                       # > 16 tensors: the framework pair stays
    c = ops.ClipAdam(o, 40.0)
    with pytest.raises(Exception):
        c.step()                                                # no gradients yet


def test_clip_adam_non_finite_gradient_norm_poisons_every_parameter_like_torch():
    """one NaN in one gradient: torch.nn.utils.clip_grad_norm_ multiplies EVERY gradient by clamp(NaN, max=1) = NaN,
    so the whole model turns NaN in that step — visible at once.  The fused kernel must not take a normal Adam
    step on the clean tensors (a partly corrupted model)."""
    from parl_amd import ops
    from parl_amd.algorithms.impala.graphed import make_capturable
    dev = torch.device('cuda', 0)
    shapes = [(16, 4, 4, 4), (16, ), (300, 17)]
    pa, pb = _models(shapes, dev, 0)
    oa, ob = torch.optim.Adam(pa, lr=1e-3), torch.optim.Adam(pb, lr=1e-3)
    make_capturable(oa, dev)
    ca = ops.ClipAdam(oa, 40.0)
    g = torch.Generator(device='cpu').manual_seed(2)
    grads = [torch.randn(s, generator=g) for s in shapes]
    grads[1][3] = float('nan')
    for p, q, gr in zip(pa, pb, grads):
        p.grad, q.grad = gr.to(dev), gr.to(dev).clone()
    ca.step()
    torch.nn.utils.clip_grad_norm_(pb, 40.0)
    ob.step()
    torch.cuda.synchronize()
    for p, q in zip(pa, pb):
        assert bool(torch.isnan(q).all()) and bool(torch.isnan(p).all())


def test_eager_clip_adam_step_is_seen_by_the_cached_actor_layouts():
    """ClipAdam.step() writes parameters through raw pointers (their autograd version never moves) while
    ops._cached_layout keys the actors' MFMA weight layouts on that version: after an eager step a no_grad forward of
    the 84x84 model must use the NEW conv2 / conv3 weights, not a layout cached before the step."""
    from parl_amd import ops
    from parl_amd.algorithms.impala.graphed import make_capturable
    from parl_amd.models import AtariModel84
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    m = AtariModel84(6).to(dev)
    obs = torch.randint(0, 256, (5, 4, 84, 84), dtype=torch.uint8, device=dev)
    with torch.no_grad():
        before = m.policy(obs).clone()   # fills the layout cache
    opt = torch.optim.Adam(m.parameters(), lr=1e-2)
    make_capturable(opt, dev)
    ca = ops.ClipAdam(opt, 40.0)
    g = torch.Generator(device='cpu').manual_seed(3)
    for p in m.parameters():
        p.grad = torch.randn(p.shape, generator=g).to(dev)
    ca.step()
    with torch.no_grad():
        after = m.policy(obs)
        ref = copy_of(m).policy(obs)     # a fresh module holding the same (updated) weights: nothing cached for it
    assert not torch.equal(before, after)
    assert torch.equal(after, ref)


def copy_of(m):
    import copy
    c = copy.deepcopy(m)
    for p in c.parameters():
        p.grad = None
    return c
