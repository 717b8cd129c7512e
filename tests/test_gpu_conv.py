"""Fused conv1+conv2 MFMA kernel of the IMPALA Atari model (parlhip_atari42_conv12_u8_f32, through
the C ABI) against a plain PyTorch fp32/fp64 reference of the same layers
(examples/IMPALA/atari_model.py:59-71).  Tolerance: 1e-5 relative to the activation scale (exact
f32 MFMA accumulation, only the summation order differs).  -m gpu."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _reference(obs_u8, w1, b1, w2, b2, dtype):
    x = obs_u8.to(dtype) / 255.0
    x = F.relu(F.conv2d(x, w1.to(dtype), b1.to(dtype), stride=2, padding=1))
    x = F.relu(F.conv2d(x, w2.to(dtype), b2.to(dtype), stride=2, padding=2))
    return x.flatten(1)


@pytest.mark.parametrize('n', [1, 37, 1024])
def test_conv12_matches_torch_reference(dev, n):
    from parl_amd import ops
    g = torch.Generator().manual_seed(n)
    obs = torch.randint(0, 256, (n, 4, 42, 42), generator=g, dtype=torch.uint8)
    w1 = torch.randn(16, 4, 4, 4, generator=g) * 0.2
    b1 = torch.randn(16, generator=g) * 0.1
    w2 = torch.randn(32, 16, 4, 4, generator=g) * 0.1
    b2 = torch.randn(32, generator=g) * 0.1
    out = ops.atari42_conv12(obs.to(dev), w1.to(dev), b1.to(dev), w2.to(dev), b2.to(dev)).cpu()
    ref64 = _reference(obs, w1, b1, w2, b2, torch.float64)
    assert out.shape == (n, 3872)
    scale = float(ref64.abs().max())
    err = float((out.double() - ref64).abs().max())
    assert err <= 1e-5 * scale, (err, scale)
    # and no worse than the fp32 CPU convolution itself
    ref32 = _reference(obs, w1, b1, w2, b2, torch.float32)
    assert err <= 4 * float((ref32.double() - ref64).abs().max()) + 1e-7 * scale


def test_conv12_edges_and_zero_obs(dev):
    """all-zero and all-255 observations (padding / border taps), bias-only output"""
    from parl_amd import ops
    g = torch.Generator().manual_seed(3)
    w1 = torch.randn(16, 4, 4, 4, generator=g) * 0.2
    b1 = torch.randn(16, generator=g)
    w2 = torch.randn(32, 16, 4, 4, generator=g) * 0.1
    b2 = torch.randn(32, generator=g)
    for fill in (0, 255):
        obs = torch.full((3, 4, 42, 42), fill, dtype=torch.uint8)
        out = ops.atari42_conv12(obs.to(dev), w1.to(dev), b1.to(dev), w2.to(dev), b2.to(dev)).cpu()
        ref = _reference(obs, w1, b1, w2, b2, torch.float64)
        np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5 * float(ref.abs().max()))
    assert ops.atari42_conv12(torch.zeros((0, 4, 42, 42), dtype=torch.uint8, device=dev), w1.to(dev), b1.to(dev),
                              w2.to(dev), b2.to(dev)).shape == (0, 3872)


def test_model_actor_path_equals_autograd_path(dev):
    """AtariModel42.policy under no_grad (fused kernel) == the autograd trunk (GEMM-lowered convs)"""
    from parl_amd.models import AtariModel42
    torch.manual_seed(0)
    model = AtariModel42(6).to(dev)
    obs = torch.randint(0, 256, (64, 4, 42, 42), dtype=torch.uint8, device=dev)
    with torch.no_grad():
        fast = model.policy(obs)
        slab = torch.empty((64, 6), device=dev)
        model.policy_into(obs, slab)
    slow = model.policy(obs)  # grad enabled -> GemmConv2d path
    np.testing.assert_allclose(fast.cpu().numpy(), slow.detach().cpu().numpy(), rtol=1e-4, atol=1e-4)
    assert torch.equal(fast, slab)


# ---- conv1 of the A2C model: the 84x84 -> 20x20 contraction (parlhip_atari84_conv1_u8_f32) ----
def _reference84(obs_u8, w1, b1, dtype):
    """examples/A2C/atari_model.py:21-104, first layer: obs / 255 -> conv 4->32 k8 s4 p1 -> ReLU"""
    return F.relu(F.conv2d(obs_u8.to(dtype) / 255.0, w1.to(dtype), b1.to(dtype), stride=4, padding=1))


@pytest.mark.parametrize('n', [1, 5, 256, 700])
def test_conv1_84_matches_torch_reference(dev, n):
    from parl_amd import ops
    g = torch.Generator().manual_seed(100 + n)
    obs = torch.randint(0, 256, (n, 4, 84, 84), generator=g, dtype=torch.uint8)
    w1 = torch.randn(32, 4, 8, 8, generator=g) * 0.1
    b1 = torch.randn(32, generator=g) * 0.1
    out = ops.atari84_conv1(obs.to(dev), w1.to(dev), b1.to(dev)).cpu()
    ref64 = _reference84(obs, w1, b1, torch.float64)
    assert out.shape == (n, 32, 20, 20)
    scale = float(ref64.abs().max())
    err = float((out.double() - ref64).abs().max())
    assert err <= 1e-5 * scale, (err, scale)
    ref32 = _reference84(obs, w1, b1, torch.float32)
    assert err <= 4 * float((ref32.double() - ref64).abs().max()) + 1e-7 * scale


def test_conv1_84_taps_and_borders(dev):
    """One-hot weights select single input taps, so the gather index arithmetic (channel, kernel row /
    column, padding row 0 / column 0, the unused last input row / column) is checked exactly; plus
    constant images and the empty batch."""
    from parl_amd import ops
    g = torch.Generator().manual_seed(9)
    obs = torch.randint(0, 256, (3, 4, 84, 84), generator=g, dtype=torch.uint8)
    b1 = torch.zeros(32)
    for rep in range(4):
        w1 = torch.zeros(32, 4, 8, 8)
        taps = []
        for o in range(32):
            c, kh, kw = int(torch.randint(0, 4, (1, ), generator=g)), (o + rep) % 8, (3 * o + rep) % 8
            w1[o, c, kh, kw] = 1.0
            taps.append((c, kh, kw))
        out = ops.atari84_conv1(obs.to(dev), w1.to(dev), b1.to(dev)).cpu()
        pad = F.pad(obs.float() / 255.0, (1, 1, 1, 1))
        for o, (c, kh, kw) in enumerate(taps):
            want = pad[:, c, kh:kh + 77:4, kw:kw + 77:4]
            assert torch.equal(out[:, o], want), (rep, o, c, kh, kw)
    w1 = torch.randn(32, 4, 8, 8, generator=g) * 0.1
    b1 = torch.randn(32, generator=g)
    for fill in (0, 255):
        o8 = torch.full((2, 4, 84, 84), fill, dtype=torch.uint8)
        out = ops.atari84_conv1(o8.to(dev), w1.to(dev), b1.to(dev)).cpu()
        ref = _reference84(o8, w1, b1, torch.float64)
        np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5 * float(ref.abs().max()))
    assert ops.atari84_conv1(torch.zeros((0, 4, 84, 84), dtype=torch.uint8, device=dev), w1.to(dev),
                             b1.to(dev)).shape == (0, 32, 20, 20)
    with pytest.raises(Exception):
        ops.atari84_conv1(torch.zeros((2, 4, 42, 42), dtype=torch.uint8, device=dev), w1.to(dev), b1.to(dev))


def test_model84_actor_path_equals_autograd_path(dev):
    """AtariModel84 under no_grad (MFMA conv1 on uint8) == the autograd trunk (GEMM-lowered conv1)"""
    from parl_amd.models import AtariModel84
    torch.manual_seed(0)
    model = AtariModel84(6).to(dev)
    obs = torch.randint(0, 256, (48, 4, 84, 84), dtype=torch.uint8, device=dev)
    with torch.no_grad():
        fp, fv = model.policy_and_value(obs)
    sp, sv = model.policy_and_value(obs)
    np.testing.assert_allclose(fp.cpu().numpy(), sp.detach().cpu().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(fv.cpu().numpy(), sv.detach().cpu().numpy(), rtol=1e-4, atol=1e-5)


# ---------------------------------------------------------------------------------------------
# the learner's side: backward of the fused conv1 + conv2 (parlhip_atari42_conv12_bwd_f32)
# ---------------------------------------------------------------------------------------------
def _ref_grads(obs, w1, b1, w2, b2, dy):
    """float64 autograd of the plain torch layers on the CPU"""
    p = [t.double().clone().requires_grad_(True) for t in (w1, b1, w2, b2)]
    x = obs.double() / 255.0
    x = F.relu(F.conv2d(x, p[0], p[1], stride=2, padding=1))
    x = F.relu(F.conv2d(x, p[2], p[3], stride=2, padding=2)).flatten(1)
    (x * dy.double()).sum().backward()
    return [t.grad for t in p]


@pytest.mark.parametrize('n', [1, 5, 300, 1100])
def test_conv12_backward_matches_fp64_autograd(dev, n):
    """n = 300 / 1100 exceed the grid (one workgroup per CU): the grid-stride accumulation and the
    fixed-order partial reduction are exercised; tolerance 1e-5 of each gradient's scale."""
    from parl_amd import ops
    g = torch.Generator().manual_seed(100 + n)
    dy = torch.randn(n, 3872, generator=g)
    if n <= 8:
        # general data.  A pre-activation within float32 rounding of zero flips its ReLU mask between
        # the f32 kernel and the f64 reference (a whole gradient term, not rounding noise; seen at
        # |z1| = 3e-8), so draw until the pre-activations keep a margin.
        for _ in range(20):
            obs = torch.randint(0, 256, (n, 4, 42, 42), generator=g, dtype=torch.uint8)
            obs[0, :, :5] = 0  # borders / dead ReLU regions
            w1 = torch.randn(16, 4, 4, 4, generator=g) * 0.2
            b1 = torch.randn(16, generator=g) * 0.1
            w2 = torch.randn(32, 16, 4, 4, generator=g) * 0.1
            b2 = torch.randn(32, generator=g) * 0.1
            z1 = F.conv2d(obs.double() / 255.0, w1.double(), b1.double(), stride=2, padding=1)
            z2 = F.conv2d(F.relu(z1), w2.double(), b2.double(), stride=2, padding=2)
            if float(z1.abs().min()) > 2e-6 and float(z2.abs().min()) > 2e-6:
                break
        else:
            pytest.fail('no tie-free draw')
    else:
        # many observations: ties cannot be avoided by luck, so make them impossible.  Pixels in
        # {0, 255}, conv1 weights multiples of 1/64 with biases odd multiples of 1/128, conv2 weights
        # multiples of 1/64 with biases odd multiples of 2^-14: every pre-activation is an exactly
        # representable non-zero dyadic number in f32 and f64 alike.
        obs = torch.randint(0, 2, (n, 4, 42, 42), generator=g, dtype=torch.uint8) * 255
        w1 = torch.randint(-12, 13, (16, 4, 4, 4), generator=g).float() / 64.0
        b1 = (2 * torch.randint(-8, 8, (16, ), generator=g).float() + 1) / 128.0
        w2 = torch.randint(-6, 7, (32, 16, 4, 4), generator=g).float() / 64.0
        b2 = (2 * torch.randint(-64, 64, (32, ), generator=g).float() + 1) / 16384.0
    a2 = ops.atari42_conv12(obs.to(dev), w1.to(dev), b1.to(dev), w2.to(dev), b2.to(dev))
    got = ops.atari42_conv12_backward(obs.to(dev), w1.to(dev), b1.to(dev), w2.to(dev), a2, dy.to(dev))
    again = ops.atari42_conv12_backward(obs.to(dev), w1.to(dev), b1.to(dev), w2.to(dev), a2, dy.to(dev))
    ref = _ref_grads(obs, w1, b1, w2, b2, dy)
    for name, a, b, r in zip(('dw1', 'db1', 'dw2', 'db2'), got, again, ref):
        assert torch.equal(a, b), name + ': not deterministic'
        scale = float(r.abs().max())
        err = float((a.cpu().double() - r).abs().max())
        assert err <= 1e-5 * scale * max(1.0, (n / 64.0) ** 0.5), (name, err, scale)


@pytest.mark.parametrize('n', [1, 5, 300, 1100])
def test_conv12_saved_activation_pair(dev, n):
    """The learner's pair of round 6 (parlhip_atari42_conv12_packed_save_u8_f32 / parlhip_atari42_conv12_bwd_saved_f32):
    the forward's `out` is bit-identical to the plain forward, its saved a1 is the zero-padded relu(conv1) tile (against
    fp64 torch, 1e-5 of scale, borders exactly zero), and the backward that reads a1 back instead of recomputing conv1
    gives the fp64 autograd gradients at the tolerance of the recompute kernel, run-to-run bit-identical, and equal to
    the recompute kernel's up to the order of two sums."""
    from parl_amd import ops
    g = torch.Generator().manual_seed(400 + n)
    dy = torch.randn(n, 3872, generator=g)
    if n <= 8:
        for _ in range(20):
            obs = torch.randint(0, 256, (n, 4, 42, 42), generator=g, dtype=torch.uint8)
            obs[0, :, :5] = 0
            w1 = torch.randn(16, 4, 4, 4, generator=g) * 0.2
            b1 = torch.randn(16, generator=g) * 0.1
            w2 = torch.randn(32, 16, 4, 4, generator=g) * 0.1
            b2 = torch.randn(32, generator=g) * 0.1
            z1 = F.conv2d(obs.double() / 255.0, w1.double(), b1.double(), stride=2, padding=1)
            z2 = F.conv2d(F.relu(z1), w2.double(), b2.double(), stride=2, padding=2)
            if float(z1.abs().min()) > 2e-6 and float(z2.abs().min()) > 2e-6:
                break
        else:
            pytest.fail('no tie-free draw')
    else:   # tie-free by construction (see test_conv12_backward_matches_fp64_autograd)
        obs = torch.randint(0, 2, (n, 4, 42, 42), generator=g, dtype=torch.uint8) * 255
        w1 = torch.randint(-12, 13, (16, 4, 4, 4), generator=g).float() / 64.0
        b1 = (2 * torch.randint(-8, 8, (16, ), generator=g).float() + 1) / 128.0
        w2 = torch.randint(-6, 7, (32, 16, 4, 4), generator=g).float() / 64.0
        b2 = (2 * torch.randint(-64, 64, (32, ), generator=g).float() + 1) / 16384.0
    d = [t.to(dev) for t in (obs, w1, b1, w2, b2)]
    pk = ops.atari42_conv12_pack(d[1], d[3])
    plain = ops.atari42_conv12(*d, packed=pk)
    a2, a1 = ops.atari42_conv12(*d, packed=pk, save_a1=True)
    assert torch.equal(a2, plain) and a1.shape == (n, 10000)
    ref1 = F.pad(F.relu(F.conv2d(obs.double() / 255.0, w1.double(), b1.double(), stride=2, padding=1)), (2, 2, 2, 2))
    t = a1.view(n, 16, 25, 25).cpu().double()
    assert float((t - ref1).abs().max()) <= 1e-5 * float(ref1.abs().max())
    border = torch.ones(25, 25, dtype=torch.bool)
    border[2:23, 2:23] = False
    assert float(t[:, :, border].abs().max()) == 0.0
    got = ops.atari42_conv12_backward(d[0], d[1], d[2], d[3], a2, dy.to(dev), packed=pk, a1=a1)
    again = ops.atari42_conv12_backward(d[0], d[1], d[2], d[3], a2, dy.to(dev), packed=pk, a1=a1)
    recompute = ops.atari42_conv12_backward(d[0], d[1], d[2], d[3], a2, dy.to(dev), packed=pk)
    ref = _ref_grads(obs, w1, b1, w2, b2, dy)
    for name, a, b, c, r in zip(('dw1', 'db1', 'dw2', 'db2'), got, again, recompute, ref):
        assert torch.equal(a, b), name + ': not deterministic'
        scale = float(r.abs().max())
        err = float((a.cpu().double() - r).abs().max())
        assert err <= 1e-5 * scale * max(1.0, (n / 64.0) ** 0.5), (name, err, scale)
        assert float((a - c).abs().max()) <= 2e-5 * scale * max(1.0, (n / 64.0) ** 0.5), name
    assert torch.equal(got[2], recompute[2]) and torch.equal(got[3], recompute[3])   # dW2 / db2: the same sums in the same order
    with pytest.raises(Exception):
        ops.atari42_conv12_backward(d[0], d[1], d[2], d[3], a2, dy.to(dev), packed=pk, a1=a1[:, :9999].contiguous())


def test_conv12_backward_one_hot_gradient_selects_single_taps(dev):
    """dy = one-hot at (o, oy, ox): dW2[o] must be exactly the a1 patch under that output (zero
    elsewhere) and db2 = e_o — checks the gather arithmetic of (2) without summation noise."""
    from parl_amd import ops
    g = torch.Generator().manual_seed(9)
    obs = torch.randint(0, 256, (1, 4, 42, 42), generator=g, dtype=torch.uint8)
    w1 = torch.randn(16, 4, 4, 4, generator=g) * 0.2
    b1 = torch.rand(16, generator=g)            # positive biases: a1 mostly alive
    w2 = torch.rand(32, 16, 4, 4, generator=g) * 0.1
    b2 = torch.rand(32, generator=g)            # a2 > 0 everywhere
    x = obs.float() / 255.0
    a1 = F.relu(F.conv2d(x, w1, b1, stride=2, padding=1))
    a1p = F.pad(a1, (2, 2, 2, 2))
    a2 = ops.atari42_conv12(obs.to(dev), w1.to(dev), b1.to(dev), w2.to(dev), b2.to(dev))
    assert bool((a2 > 0).all())
    for (o, oy, ox) in [(0, 0, 0), (31, 10, 10), (7, 3, 9), (16, 10, 0)]:
        dy = torch.zeros(1, 32, 11, 11)
        dy[0, o, oy, ox] = 1.0
        dw1, db1, dw2, db2 = ops.atari42_conv12_backward(obs.to(dev), w1.to(dev), b1.to(dev), w2.to(dev), a2,
                                                         dy.reshape(1, 3872).to(dev))
        patch = a1p[0, :, 2 * oy:2 * oy + 4, 2 * ox:2 * ox + 4]
        exp = torch.zeros(32, 16, 4, 4)
        exp[o] = patch
        np.testing.assert_allclose(dw2.cpu().numpy(), exp.numpy(), rtol=1e-6, atol=1e-6)
        e = torch.zeros(32)
        e[o] = 1.0
        assert torch.equal(db2.cpu(), e)


def test_model_learner_path_gradients_match_gemm_lowered_autograd(dev):
    """AtariModel42 under autograd on uint8 observations (fused forward + the backward kernel)
    vs the same parameters through the GEMM-lowered convolutions on float observations."""
    from parl_amd.models import AtariModel42
    torch.manual_seed(1)
    m = AtariModel42(6).to(dev)
    obs = torch.randint(0, 256, (96, 4, 42, 42), dtype=torch.uint8, device=dev)
    wgt = torch.randn(96, 6, device=dev)

    def grads(o):
        m.zero_grad(set_to_none=True)
        logits, v = m.policy_and_value(o)
        ((logits * wgt).sum() + (v * v).sum()).backward()
        return [p.grad.clone() for p in m.parameters()]

    ga = grads(obs)
    gb = grads(obs.float())
    for (name, _), a, b in zip(m.named_parameters(), ga, gb):
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 2e-4 * scale + 1e-6, name


# ---------------------------------------------------------------------------------------------
# the 84x84 model: fused conv2 + conv3 (parlhip_atari84_conv23_f32)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('n', [1, 7, 300])
def test_conv23_84_matches_torch_reference(dev, n):
    from parl_amd import ops
    g = torch.Generator().manual_seed(50 + n)
    a1 = torch.relu(torch.randn(n, 32, 20, 20, generator=g))
    w2 = torch.randn(64, 32, 4, 4, generator=g) * 0.05
    b2 = torch.randn(64, generator=g) * 0.1
    w3 = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    b3 = torch.randn(64, generator=g) * 0.1
    a3, a2 = ops.atari84_conv23(a1.to(dev), w2.to(dev), b2.to(dev), w3.to(dev), b3.to(dev), save_a2=True)
    r2 = F.relu(F.conv2d(a1.double(), w2.double(), b2.double(), stride=2, padding=2))
    r3 = F.relu(F.conv2d(r2, w3.double(), b3.double())).flatten(1)
    for got, ref in ((a2.cpu().double(), r2), (a3.cpu().double(), r3)):
        assert got.shape == ref.shape
        assert float((got - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    only3 = ops.atari84_conv23(a1.to(dev), w2.to(dev), b2.to(dev), w3.to(dev), b3.to(dev))
    assert torch.equal(only3, a3)


def test_conv23_84_one_hot_weights_select_single_taps(dev):
    """one-hot conv2 / conv3 weights: the output must be exactly one (shifted, strided) input plane —
    checks the streamed-weight layout (wt2 / wt3) and the gather arithmetic without summation noise"""
    from parl_amd import ops
    g = torch.Generator().manual_seed(4)
    a1 = torch.rand(2, 32, 20, 20, generator=g)
    for (o2, c2, kh2, kw2, o3, kh3, kw3) in [(0, 0, 0, 0, 0, 0, 0), (63, 31, 3, 3, 63, 2, 2), (17, 5, 1, 2, 40, 1, 0), (33, 20, 2, 0, 9, 0, 2)]:
        w2 = torch.zeros(64, 32, 4, 4)
        w2[o2, c2, kh2, kw2] = 1.0
        w3 = torch.zeros(64, 64, 3, 3)
        w3[o3, o2, kh3, kw3] = 1.0
        z = torch.zeros(64)
        a3, a2 = ops.atari84_conv23(a1.to(dev), w2.to(dev), z.to(dev), w3.to(dev), z.to(dev), save_a2=True)
        r2 = F.conv2d(a1, w2, None, stride=2, padding=2)
        r3 = F.conv2d(r2, w3).flatten(1)
        assert torch.equal(a2.cpu(), r2) and torch.equal(a3.cpu(), r3)


def _dyadic(gen, shape, lo, hi, denom):
    return torch.randint(lo, hi, shape, generator=gen).float() / denom


@pytest.mark.parametrize('n', [1, 3, 300])
def test_conv3_84_backward_matches_fp64_autograd(dev, n):
    """dyadic data (activations k/8, weights k/32, biases odd/512): every pre-activation is an exactly
    representable non-zero number, so the ReLU masks of the f32 kernels and the f64 reference agree"""
    from parl_amd import ops
    g = torch.Generator().manual_seed(70 + n)
    a2 = torch.relu(_dyadic(g, (n, 64, 11, 11), -8, 9, 8.0))
    w3 = _dyadic(g, (64, 64, 3, 3), -4, 5, 32.0)
    b3 = (2 * torch.randint(-16, 16, (64, ), generator=g).float() + 1) / 512.0
    dy = torch.randn(n, 5184, generator=g)
    p = [t.double().clone().requires_grad_(True) for t in (a2, w3, b3)]
    a3r = F.relu(F.conv2d(p[0], p[1], p[2])).flatten(1)
    (a3r * dy.double()).sum().backward()
    a3 = a3r.detach().float()
    dz2, dw3, db3 = ops.atari84_conv3_backward(a2.to(dev), a3.to(dev), dy.to(dev), w3.to(dev))
    again = ops.atari84_conv3_backward(a2.to(dev), a3.to(dev), dy.to(dev), w3.to(dev))
    ref_dz2 = p[0].grad * (a2 > 0).double()
    for name, got, got2, ref in (('dz2', dz2, again[0], ref_dz2), ('dw3', dw3, again[1], p[1].grad), ('db3', db3, again[2], p[2].grad)):
        assert torch.equal(got, got2), name
        scale = float(ref.abs().max())
        assert float((got.cpu().double() - ref).abs().max()) <= 1e-5 * scale * max(1.0, (n / 64.0) ** 0.5), name


@pytest.mark.parametrize('n', [1, 3, 300])
def test_conv2_and_conv1_84_backward_match_fp64_autograd(dev, n):
    """conv2 backward (dW2, db2, dz1) and conv1 backward (dW1, db1) on tie-free dyadic data: binary
    pixels, conv1 weights k/64 with biases odd/128 (a1 exact, non-zero pre-activations)"""
    from parl_amd import ops
    g = torch.Generator().manual_seed(90 + n)
    obs = torch.randint(0, 2, (n, 4, 84, 84), generator=g, dtype=torch.uint8) * 255
    w1 = _dyadic(g, (32, 4, 8, 8), -6, 7, 64.0)
    b1 = (2 * torch.randint(-8, 8, (32, ), generator=g).float() + 1) / 128.0
    w2 = _dyadic(g, (64, 32, 4, 4), -4, 5, 32.0)
    dz2 = torch.randn(n, 64, 11, 11, generator=g)
    p = [t.double().clone().requires_grad_(True) for t in (w1, b1, w2)]
    a1r = F.relu(F.conv2d(obs.double() / 255.0, p[0], p[1], stride=4, padding=1))
    z2 = F.conv2d(a1r, p[2], None, stride=2, padding=2)
    a1r.retain_grad()
    (z2 * dz2.double()).sum().backward()
    a1 = ops.atari84_conv1(obs.to(dev), w1.to(dev), b1.to(dev))
    assert float((a1.cpu().double() - a1r.detach()).abs().max()) <= 1e-6
    dz1, dw2, db2 = ops.atari84_conv2_backward(a1, dz2.to(dev), w2.to(dev))
    ref_dz1 = a1r.grad * (a1r.detach() > 0).double()
    dw1, db1 = ops.atari84_conv1_backward(obs.to(dev), dz1)
    tol = 1e-5 * max(1.0, (n / 64.0) ** 0.5)
    for name, got, ref in (('dz1', dz1, ref_dz1), ('dw2', dw2, p[2].grad), ('db2', db2, dz2.double().sum((0, 2, 3))),
                           ('dw1', dw1, p[0].grad), ('db1', db1, p[1].grad)):
        scale = float(ref.abs().max())
        assert float((got.cpu().double() - ref).abs().max()) <= tol * scale, name
    again = ops.atari84_conv2_backward(a1, dz2.to(dev), w2.to(dev))
    assert all(torch.equal(a, b) for a, b in zip((dz1, dw2, db2), again))


def test_model84_learner_path_gradients_match_gemm_lowered_autograd(dev):
    """AtariModel84 under autograd on uint8 observations (MFMA forward + the three backward kernels)
    vs the same parameters through the GEMM-lowered convolutions on float observations"""
    from parl_amd.models import AtariModel84
    torch.manual_seed(2)
    m = AtariModel84(6).to(dev)
    obs = torch.randint(0, 256, (40, 4, 84, 84), dtype=torch.uint8, device=dev)
    wgt = torch.randn(40, 6, device=dev)

    def grads(o):
        m.zero_grad(set_to_none=True)
        logits, v = m.policy_and_value(o)
        ((logits * wgt).sum() + (v * v).sum()).backward()
        return [p.grad.clone() for p in m.parameters()]

    ga, gb = grads(obs), grads(obs.float())
    for (name, _), a, b in zip(m.named_parameters(), ga, gb):
        assert float((a - b).abs().max()) <= 3e-4 * float(b.abs().max()) + 1e-6, name


def test_conv12_packed_weights_give_identical_outputs(dev):
    """parlhip_atari42_conv12_weights_f32 + the _packed_ forward entries: the operand-order copy of the weights is
    data movement only — outputs bit-identical to the nn.Conv2d-layout entries, plain and ring-reading"""
    from parl_amd import ops
    from parl_amd.env import DeviceVectorEnv
    torch.manual_seed(5)
    w1, b1 = torch.randn(16, 4, 4, 4, device=dev) * 0.2, torch.randn(16, device=dev) * 0.1
    w2, b2 = torch.randn(32, 16, 4, 4, device=dev) * 0.1, torch.randn(32, device=dev) * 0.1
    pk = ops.atari42_conv12_pack(w1, w2)
    assert pk.numel() == 36 * 64 * 4 + 4 * 8 * 64 * 4
    # every weight appears exactly once in the forward region, every conv2 weight once more in the backward region
    assert torch.equal(torch.sort(pk[:9216]).values, torch.sort(torch.cat([w1.flatten(), w2.flatten()])).values)
    assert torch.equal(torch.sort(pk[9216:]).values, torch.sort(w2.flatten()).values)
    for n in (3, 300, 1100):   # the backward with the operand-order weights: bit-identical gradients
        obs = torch.randint(0, 256, (n, 4, 42, 42), dtype=torch.uint8, device=dev)
        a2 = ops.atari42_conv12(obs, w1, b1, w2, b2)
        dy = torch.randn_like(a2)
        for x, y in zip(ops.atari42_conv12_backward(obs, w1, b1, w2, a2, dy),
                        ops.atari42_conv12_backward(obs, w1, b1, w2, a2, dy, packed=pk)):
            assert torch.equal(x, y)
    for n in (1, 3, 700, 1030):
        obs = torch.randint(0, 256, (n, 4, 42, 42), dtype=torch.uint8, device=dev)
        assert torch.equal(ops.atari42_conv12(obs, w1, b1, w2, b2), ops.atari42_conv12(obs, w1, b1, w2, b2, packed=pk))
    try:
        env = DeviceVectorEnv('PongNoFrameskip-v4', 9, dim=42, horizon=8, seed=2, device=dev)
    except FileNotFoundError:
        pytest.skip('cartridge pong.bin not present')
    env.reset()
    for i in range(5):
        env.step(torch.randint(0, 6, (9, ), device=dev))
        ref = env.current_obs_ref()
        assert torch.equal(ops.atari42_conv12(ref, w1, b1, w2, b2), ops.atari42_conv12(ref, w1, b1, w2, b2, packed=pk))


def test_model42_packed_weights_follow_the_weights(dev):
    """AtariModel42 keeps ONE operand-order buffer (fixed address: hipGraph segments read it); it is rebuilt when
    the parameters' version counters moved, after set_weights / sync_weights_to, and on refresh_actor_layout()"""
    from parl_amd.models import AtariModel42
    torch.manual_seed(1)
    m, other = AtariModel42(6).to(dev), AtariModel42(6).to(dev)
    obs = torch.randint(0, 256, (33, 4, 42, 42), dtype=torch.uint8, device=dev)

    def check():
        with torch.no_grad():
            got = m.policy_hidden(obs)
            h = ops.atari42_conv12(obs, m.conv1.weight, m.conv1.bias, m.conv2.weight, m.conv2.bias)   # unpacked
            want = torch._addmm_activation(m.conv3.bias, h, m.conv3.weight.flatten(1).t(), use_gelu=False)
        assert torch.equal(got, want)

    from parl_amd import ops
    check()
    addr = m._wpk.data_ptr()
    with torch.no_grad():
        m.conv2.weight.mul_(1.5)                      # an eager in-place write: the version counter moves
    check()
    with torch.no_grad():
        torch._foreach_copy_(list(m.parameters()), list(other.parameters()))   # the actors' snapshot copy
    check()
    other.conv1.weight.data.add_(0.25)
    other.sync_weights_to(m)                          # .data writes: no version bump, the model refreshes itself
    check()
    m.set_weights(AtariModel42(6).get_weights())
    check()
    m.conv1.weight.data.mul_(2.0)                     # nobody told the model ...
    m.refresh_actor_layout()                          # ... until now
    check()
    assert m._wpk.data_ptr() == addr


def test_conv1_84_operand_order_weights_give_identical_outputs(dev):
    """parlhip_atari84_conv1_[ring_]packed_u8_f32 (what ops.atari84_conv1 calls) against the nn.Conv2d-layout entries"""
    from parl_amd import _native as N
    from parl_amd import ops
    torch.manual_seed(9)
    w1, b1 = torch.randn(32, 4, 8, 8, device=dev) * 0.1, torch.randn(32, device=dev) * 0.1
    for n in (1, 5, 600):
        obs = torch.randint(0, 256, (n, 4, 84, 84), dtype=torch.uint8, device=dev)
        want = torch.empty((n, 32, 20, 20), dtype=torch.float32, device=dev)
        N.check(N.lib().parlhip_atari84_conv1_u8_f32(N.ptr(obs), N.ptr(w1), N.ptr(b1), N.ptr(want), n, N.stream_ptr()), 'conv1_84')
        assert torch.equal(ops.atari84_conv1(obs, w1, b1), want)
    w1.mul_(0.5)   # the cached operand-order copy follows the tensor's version
    N.check(N.lib().parlhip_atari84_conv1_u8_f32(N.ptr(obs), N.ptr(w1), N.ptr(b1), N.ptr(want), n, N.stream_ptr()), 'conv1_84')
    assert torch.equal(ops.atari84_conv1(obs, w1, b1), want)
