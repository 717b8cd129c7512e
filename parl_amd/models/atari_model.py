"""The two Atari networks of the reference examples, as torch parl.Models.

AtariModel42 — examples/IMPALA/atari_model.py:21-90 (42x42 input, 1.00 M params):
    conv 4->16 k4 s2 p1 (42->21), 16->32 k4 s2 p2 (->11), 32->256 k11 (->1), fc 256->A / 256->1
    with Normal(0,1) initialised heads (atari_model.py:44-57), obs / 255 (:66).
AtariModel84 — examples/A2C/atari_model.py:21-104 (84x84 input, 2.74 M params):
    conv 4->32 k8 s4 p1 (84->20), 32->64 k4 s2 p2 (->11), 64->64 k3 (->9), fc 5184->512,
    512->A, 512->1.
obs may arrive as uint8 straight from the rollout ring: the /255 happens after the cast on the
GPU, so the batch crosses HBM as bytes (4x less than the reference's float32 obs).

Convolutions run as im2col + one rocBLAS GEMM (`GemmConv2d`), not through MIOpen: this image
ships no precompiled MIOpen kernel database for gfx950, so every new (shape, direction) pair
would JIT-compile for minutes on a fresh box (measured: the 1024-env bench did not finish its
first learner update in 300 s).  Parameters keep nn.Conv2d's names/shapes, so state_dicts are
interchangeable with a stock torch model (and with the reference's checkpoints)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..core import Model

__all__ = ['AtariModel42', 'AtariModel84', 'GemmConv2d']


class _ColGemmFn(torch.autograd.Function):
    """out = col @ w^T + b with a backward whose weight gradient is a SPLIT-K batched GEMM.
    dW = d_out^T @ col has a tiny output ([O, K]: 32x256 ... 64x576) and an enormous reduction
    dimension (rows x positions: 2.5 M for a 6,400-row chunk of the 84x84 model); rocBLAS runs that
    as ONE column of workgroups (no split-K: 3.9 ms per call, 220 ms per 51,200-row update).  Cutting
    the reduction into S slices makes it S well-shaped GEMMs + a sum over S."""

    @staticmethod
    def forward(ctx, col, wmat, bias):
        ctx.save_for_backward(col, wmat)
        ctx.has_bias = bias is not None
        return col @ wmat.t() if bias is None else torch.addmm(bias, col, wmat.t())

    @staticmethod
    def backward(ctx, d_out):
        col, wmat = ctx.saved_tensors
        d_out = d_out.contiguous()
        rows = col.shape[0]
        d_col = d_out @ wmat if ctx.needs_input_grad[0] else None
        S = 1
        for cand in (256, 128, 64, 32, 16, 8, 4, 2):
            if rows % cand == 0 and rows // cand >= 2048:
                S = cand
                break
        if S == 1:
            d_w = d_out.t() @ col
        else:
            d_w = torch.bmm(d_out.view(S, rows // S, -1).transpose(1, 2), col.view(S, rows // S, -1)).sum(0)
        d_b = d_out.sum(0) if ctx.has_bias else None
        return d_col, d_w, d_b


class GemmConv2d(nn.Conv2d):
    """nn.Conv2d whose forward is ONE strided-gather copy (im2col through a 6-D unfold view) + ONE
    rocBLAS GEMM [N*Ho*Wo, C*kh*kw] x [C*kh*kw, O]; fp32, same arithmetic up to the GEMM's
    reduction tree.  (torch's F.unfold launches one im2col kernel PER SAMPLE — 614,400 launches
    per learner update at N=51,200 — so it is not used.)  A kernel that covers the whole input is
    a plain linear layer.  The output is an NCHW-shaped view of the [N, Ho, Wo, O] GEMM result."""

    def forward(self, x):
        kh, kw = self.kernel_size
        n, c, h, w = x.shape
        wmat = self.weight.flatten(1)  # [O, C*kh*kw]
        if (h, w) == (kh, kw) and self.padding == (0, 0):
            return F.linear(x.flatten(1), wmat, self.bias).view(n, -1, 1, 1)
        ph, pw = self.padding
        if ph or pw:
            x = F.pad(x, (pw, pw, ph, ph))
        patches = x.unfold(2, kh, self.stride[0]).unfold(3, kw, self.stride[1])  # [N,C,Ho,Wo,kh,kw] view
        ho, wo = patches.shape[2], patches.shape[3]
        col = patches.permute(0, 2, 3, 1, 4, 5).reshape(n * ho * wo, c * kh * kw)
        out = _ColGemmFn.apply(col, wmat, self.bias)
        return out.view(n, ho, wo, -1).permute(0, 3, 1, 2)


class AtariModel42(Model):
    def __init__(self, act_dim):
        super(AtariModel42, self).__init__()
        self.conv1 = GemmConv2d(4, 16, kernel_size=4, stride=2, padding=1)
        self.conv2 = GemmConv2d(16, 32, kernel_size=4, stride=2, padding=2)
        self.conv3 = GemmConv2d(32, 256, kernel_size=11, stride=1, padding=0)
        self.policy_fc = nn.Linear(256, act_dim)
        self.value_fc = nn.Linear(256, 1)
        # (the three convolutions keep torch's default initialisation — the reference's take Paddle's, see
        # paddle_default_init_ below; this model's learning curves (tests/test_gpu_learning.py, profiles/) were all
        # measured with torch's and pass the reference's published ones, so it was left alone)
        for fc in (self.policy_fc, self.value_fc):  # paddle Normal() initializer: N(0, 1)
            nn.init.normal_(fc.weight, 0.0, 1.0)
            nn.init.normal_(fc.bias, 0.0, 1.0)

    reads_ring = True   # the actors' step may hand _trunk an ops.RingObservation (DeviceRollout.collect_step)

    # The actors' conv1 / conv2 weights in the MFMA kernel's operand order (ops.atari42_conv12_pack): ONE buffer at a
    # fixed address for the life of the model — a rollout segment replayed as a hipGraph reads it — rebuilt when the
    # weights were written: by whoever copies new weights in (refresh_actor_layout(): AsyncActorLearner after every
    # snapshot / mid-rollout refresh, on the stream of the copy, so the events that order the weights order it too),
    # or here, in an eager call, when the parameters' version counters moved.
    _wpk, _wpk_key = None, None

    def _packed_weights(self, force=False):
        w1, w2 = self.conv1.weight, self.conv2.weight
        if self._wpk is None or self._wpk.device != w1.device:
            self._wpk, self._wpk_key = ops.atari42_conv12_pack(w1, w2), None
        key = (w1._version, w2._version, w1.data_ptr(), w2.data_ptr())
        written = getattr(w1, '_parl_graph_written', False) or getattr(w2, '_parl_graph_written', False)
        if force or written or key != self._wpk_key:   # (parameters a graph replay / raw-pointer optimizer writes: never trusted)
            ops.atari42_conv12_pack(w1, w2, out=self._wpk)
            self._wpk_key = key
        return self._wpk

    def refresh_actor_layout(self):
        """call after writing conv1 / conv2 weights in a way the version counters may not see, or to put the rebuild
        on the stream (and in front of the events) of a weight copy"""
        if self.conv1.weight.is_cuda:
            self._packed_weights(force=True)

    def _trunk(self, obs):
        if isinstance(obs, ops.RingObservation) and (torch.is_grad_enabled() or obs.dim != 42 or obs.shape[0] == 0):
            obs = obs.materialize()
        if (not torch.is_grad_enabled()) and obs.dtype == torch.uint8 and obs.is_cuda and obs.shape[0] > 0:
            # the actors' path (no autograd): conv1+conv2 as one fused MFMA kernel on the uint8
            # observations (ops.atari42_conv12; a RingObservation is read in place), conv3 is a 3872 -> 256 linear layer
            h = ops.atari42_conv12(obs, self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias,
                                   packed=self._packed_weights())
            # conv3 + ReLU as ONE GEMM with a ReLU epilogue (hipBLASLt) instead of addmm + clamp
            return torch._addmm_activation(self.conv3.bias, h, self.conv3.weight.flatten(1).t(), use_gelu=False)
        if obs.dtype == torch.uint8 and obs.is_cuda and obs.shape[0] > 0:
            # the learner's path: the same fused forward kernel under autograd, its backward is ONE
            # kernel that recomputes conv1 and produces the four parameter gradients (no im2col, no
            # stored conv1 activation); conv3 = a 3872 -> 256 linear layer = plain rocBLAS GEMMs
            h = ops.Atari42Conv12Fn.apply(obs, self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias)
            return F.relu(F.linear(h, self.conv3.weight.flatten(1), self.conv3.bias))
        x = obs.float() / 255.0
        x = F.relu(self.conv1(x))
        x = F.relu(self.conv2(x))
        x = F.relu(self.conv3(x))
        return x.flatten(1)

    def policy(self, obs):
        return self.policy_fc(self._trunk(obs))

    @torch.no_grad()
    def policy_into(self, obs, out):
        """policy(obs) written straight into `out` (a [E, A] slab of a rollout buffer)"""
        return torch.addmm(self.policy_fc.bias, self._trunk(obs), self.policy_fc.weight.t(), out=out)

    supports_offset_base = True   # policy_sample_into(offset_base=...): the rollout may be replayed as a hipGraph

    @torch.no_grad()
    def policy_sample_into(self, obs, logits_out, actions_out, seed, offset, row0=0, offset_base=None):
        """the actors' step: policy(obs) into `logits_out` AND the sampled actions into `actions_out` (slabs of
        a rollout buffer), head + draw in one launch (ops.policy_head_sample_into).  offset_base: int64 [1] device
        tensor added to `offset` on the device (DeviceRollout keeps the number of the rollout's first step there)."""
        h = self._trunk(obs)
        if not ops.policy_head_sample_into(h, self.policy_fc.weight, self.policy_fc.bias, logits_out, actions_out,
                                           seed, offset, row0, offset_base):
            if offset_base is not None:
                raise RuntimeError('policy_sample_into: no fused head + draw for this shape, offset_base unsupported')
            torch.addmm(self.policy_fc.bias, h, self.policy_fc.weight.t(), out=logits_out)
            ops.policy_sample_into(logits_out, actions_out, seed, offset, row0)

    @torch.no_grad()
    def policy_hidden(self, obs):
        """the trunk output [E, 256] for a consumer that runs policy_fc + the draw itself: the device env's step
        (DeviceVectorEnv.step_policy_async — head, draw, emulator and observation in one launch)"""
        return self._trunk(obs)

    def value(self, obs):
        return self.value_fc(self._trunk(obs)).squeeze(1)

    def policy_and_value(self, obs):
        h = self._trunk(obs)
        return self.policy_fc(h), self.value_fc(h).squeeze(1)


def paddle_default_init_(module):
    """The initialisation the reference's model gets: examples/A2C/atari_model.py:21-104 names no initializer, so its
    layers take PADDLE's defaults — nn.Conv2D: weight ~ Normal(0, sqrt(2 / (k_h k_w in_channels))) (He), nn.Linear:
    weight ~ Xavier uniform U(+-sqrt(6 / (fan_in + fan_out))), every bias 0.  torch's own defaults
    (kaiming_uniform(a=sqrt(5)): U(+-1/sqrt(fan_in)) for weights AND biases) are 2.3-2.4x smaller per layer; four
    layers deep that leaves pre-activations so small that the first Adam steps at the reference's lr = 1e-3 — each
    moves every weight by the full learning rate — can switch off every ReLU of the 512-unit layer for good:
    with torch's defaults `examples/A2C/train.py --seed 1` never left the uniform policy (critic loss flat at the 425
    of a constant prediction; profiles/README.md, the r04 A2C rows)."""
    with torch.no_grad():
        for m in module.modules():
            if isinstance(m, nn.Conv2d):
                fan_in = m.in_channels * m.kernel_size[0] * m.kernel_size[1]
                m.weight.normal_(0.0, (2.0 / fan_in) ** 0.5)
                if m.bias is not None:
                    m.bias.zero_()
            elif isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    m.bias.zero_()
    return module


class AtariModel84(Model):
    def __init__(self, act_dim):
        super(AtariModel84, self).__init__()
        self.conv1 = GemmConv2d(4, 32, kernel_size=8, stride=4, padding=1)
        self.conv2 = GemmConv2d(32, 64, kernel_size=4, stride=2, padding=2)
        self.conv3 = GemmConv2d(64, 64, kernel_size=3, stride=1, padding=0)
        self.fc = nn.Linear(5184, 512)
        self.policy_fc = nn.Linear(512, act_dim)
        self.value_fc = nn.Linear(512, 1)
        paddle_default_init_(self)

    reads_ring = True   # the actors' step may hand _trunk an ops.RingObservation (DeviceRollout / DeviceA2CRollout)

    # The three convolutions' weights in MFMA operand order for the no-grad (actors' / bootstrap-value) path: ONE set of
    # buffers at fixed addresses for the life of the model, so that a rollout captured as a hipGraph
    # (DeviceA2CRollout) reads them; rebuilt when the parameters' version counters moved (an eager call), or by
    # refresh_actor_layout() — which a graphed rollout calls INSIDE its graph, so every replay starts from the current
    # weights — and while `_lay_pinned` is set nothing is checked.
    _lay, _lay_key, _lay_pinned = None, None, False

    def _actor_layouts(self, force=False):
        ws = (self.conv1.weight, self.conv2.weight, self.conv3.weight)
        if self._lay is None or self._lay[0].device != ws[0].device:
            with torch.no_grad():
                self._lay = (ops.atari84_conv1_layout(ws[0]), ) + tuple(ops.atari84_conv23_layouts(ws[1], ws[2]))
            self._lay_key = None
        if self._lay_pinned and not force:
            return self._lay
        key = tuple((w._version, w.data_ptr()) for w in ws)
        written = any(getattr(w, '_parl_graph_written', False) for w in ws)
        if force or written or key != self._lay_key:
            with torch.no_grad():
                self._lay[0].copy_(ops.atari84_conv1_layout(ws[0]))
                wt2, wt3 = ops.atari84_conv23_layouts(ws[1], ws[2])
                self._lay[1].copy_(wt2)
                self._lay[2].copy_(wt3)
            self._lay_key = key
        return self._lay

    def refresh_actor_layout(self):
        if self.conv1.weight.is_cuda:
            self._actor_layouts(force=True)

    def _trunk(self, obs):
        if isinstance(obs, ops.RingObservation) and (torch.is_grad_enabled() or obs.dim != 84 or obs.shape[0] == 0):
            obs = obs.materialize()
        if (not torch.is_grad_enabled()) and obs.dtype == torch.uint8 and obs.is_cuda and obs.shape[0] > 0:
            # the actors' / bootstrap-value path (no autograd): the 84x84 -> 20x20 contraction as one
            # MFMA kernel on the uint8 observations with /255, bias and ReLU fused (ops.atari84_conv1)
            wt1, wt2, wt3 = self._actor_layouts()
            x = ops.atari84_conv1(obs, self.conv1.weight, self.conv1.bias, wt1=wt1)
            # conv2 + conv3 fused (a2 stays in LDS, weights streamed from L2 in MFMA operand order)
            x = ops.atari84_conv23(x, self.conv2.weight, self.conv2.bias, self.conv3.weight, self.conv3.bias,
                                   wt23=(wt2, wt3))
            return F.relu(self.fc(x))
        if obs.dtype == torch.uint8 and obs.is_cuda and obs.shape[0] > 0:
            # the learner's path: the same forward kernels under ONE autograd node whose backward is three
            # per-layer MFMA kernels (ops.Atari84TrunkFn): no im2col; fc and the heads stay rocBLAS GEMMs
            x = ops.Atari84TrunkFn.apply(obs, self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias,
                                         self.conv3.weight, self.conv3.bias)
            return F.relu(self.fc(x))
        x = F.relu(self.conv1(obs.float() / 255.0))
        x = F.relu(self.conv2(x))
        x = F.relu(self.conv3(x))
        return F.relu(self.fc(x.flatten(1)))

    def policy(self, obs):
        return self.policy_fc(self._trunk(obs))

    def value(self, obs):
        return self.value_fc(self._trunk(obs)).squeeze(1)

    def policy_and_value(self, obs):
        h = self._trunk(obs)
        return self.policy_fc(h), self.value_fc(h).squeeze(1)
