"""IMPALA for the torch host framework with the API of the reference's Paddle implementation
(parl/algorithms/paddle/impala/impala.py:25-227) — the reference ships no torch IMPALA
(parl/algorithms/torch/__init__.py:15-29), so this is the torch twin.

What runs where:
  * model forward/backward, log-softmax gather, entropy, KL, Adam, global-norm clip: PyTorch-ROCm;
  * _log_prob of the behaviour policy, discounts, the [B*T]->[T-1,B] slicing and V-trace:
    ONE fused HIP kernel (ops.vtrace_from_logits), no Python loop over time.
The value and policy heads share one trunk pass when the model offers `policy_and_value`
(mathematically identical to the reference's two passes, impala.py:148-149).
"""
import torch
import torch.nn.functional as F

from ... import ops
from ...core import Algorithm
from . import vtrace

__all__ = ['IMPALA', 'VTraceLoss']


class VTraceLoss(object):
    """Same constructor arguments and attributes as impala.py:25-79; all tensors [T,B]."""

    def __init__(self,
                 behaviour_actions_log_probs,
                 target_actions_log_probs,
                 policy_entropy,
                 dones,
                 discount,
                 rewards,
                 values,
                 bootstrap_value,
                 entropy_coeff=-0.01,
                 vf_loss_coeff=0.5,
                 clip_rho_threshold=1.0,
                 clip_pg_rho_threshold=1.0):
        self.vtrace_returns = vtrace.from_importance_weights(
            behaviour_actions_log_probs=behaviour_actions_log_probs.detach(),
            target_actions_log_probs=target_actions_log_probs.detach(),
            discounts=(~dones).float() * discount,
            rewards=rewards,
            values=values.detach(),
            bootstrap_value=bootstrap_value.detach(),
            clip_rho_threshold=clip_rho_threshold,
            clip_pg_rho_threshold=clip_pg_rho_threshold)
        _finish_loss(self, target_actions_log_probs, values, policy_entropy, entropy_coeff, vf_loss_coeff)


def _finish_loss(self, target_actions_log_probs, values, policy_entropy, entropy_coeff, vf_loss_coeff):
    # sums, not means (impala.py:67-79)
    self.pi_loss = -1.0 * torch.sum(target_actions_log_probs * self.vtrace_returns.pg_advantages)
    delta = values - self.vtrace_returns.vs
    self.vf_loss = 0.5 * torch.sum(torch.square(delta))
    self.entropy = torch.sum(policy_entropy)
    self.total_loss = self.pi_loss + self.vf_loss * vf_loss_coeff + self.entropy * entropy_coeff


class _FusedVTraceLoss(object):
    """VTraceLoss whose V-trace targets came from the fused from-logits kernel."""

    def __init__(self, vs, pg_adv, target_actions_log_probs, values, policy_entropy, entropy_coeff,
                 vf_loss_coeff):
        self.vtrace_returns = vtrace.VTraceReturns(vs=vs, pg_advantages=pg_adv)
        _finish_loss(self, target_actions_log_probs, values, policy_entropy, entropy_coeff, vf_loss_coeff)


class _FusedLossFn(torch.autograd.Function):
    """total_loss of VTraceLoss computed, together with its gradient w.r.t. (target_logits, values),
    by ONE kernel (ops.impala_loss); backward is two multiplies by the incoming scalar gradient."""

    @staticmethod
    def forward(ctx, target_logits, values, behaviour_logits, actions, rewards, dones, cfg):
        gamma, crho, cpg, vf_c, ent_c, time_major = cfg
        out = ops.impala_loss(behaviour_logits, target_logits, actions, rewards, dones, values, gamma, crho, cpg,
                              vf_c, ent_c, time_major=time_major)
        if out is None:
            raise NotImplementedError
        vs, pg, glog, gval, sums = out
        ctx.save_for_backward(glog, gval)
        total = (sums[0] + vf_c * sums[1] + ent_c * sums[2]).float()
        ctx.mark_non_differentiable(sums, vs, pg)
        return total, sums, vs, pg

    @staticmethod
    def backward(ctx, g_total, g_sums, g_vs, g_pg):
        glog, gval = ctx.saved_tensors
        return g_total * glog, g_total * gval, None, None, None, None, None


class _FusedHeadsLossFn(torch.autograd.Function):
    """policy_fc + value_fc + total_loss + the gradient w.r.t. the trunk output and the heads' parameters
    from ONE kernel (ops.impala_heads_loss): the trunk output crosses HBM once forward, its gradient once."""

    @staticmethod
    def forward(ctx, hidden, w_pi, b_pi, w_v, b_v, behaviour_logits, actions, rewards, dones, cfg):
        gamma, crho, cpg, vf_c, ent_c = cfg
        out = ops.impala_heads_loss(hidden, w_pi, b_pi, w_v, b_v, behaviour_logits, actions, rewards, dones, gamma,
                                    crho, cpg, vf_c, ent_c)
        if out is None:
            raise NotImplementedError
        vs, pg, gh, gwp, gbp, gwv, gbv, sums = out
        ctx.save_for_backward(gh, gwp, gbp, gwv.reshape(w_v.shape), gbv.reshape(b_v.shape))
        total = (sums[0] + vf_c * sums[1] + ent_c * sums[2]).float()
        ctx.mark_non_differentiable(sums, vs, pg)
        return total, sums, vs, pg

    @staticmethod
    def backward(ctx, g_total, g_sums, g_vs, g_pg):
        gh, gwp, gbp, gwv, gbv = ctx.saved_tensors
        return g_total * gh, g_total * gwp, g_total * gbp, g_total * gwv, g_total * gbv, None, None, None, None, None


class _KernelVTraceLoss(object):
    """VTraceLoss whose terms and gradient came from the fused loss kernel"""

    def __init__(self, total, sums, vs, pg):
        self.vtrace_returns = vtrace.VTraceReturns(vs=vs, pg_advantages=pg)
        self.total_loss = total
        self.pi_loss, self.vf_loss, self.entropy = sums[0].float(), sums[1].float(), sums[2].float()


class IMPALA(Algorithm):
    def __init__(self,
                 model,
                 sample_batch_steps=None,
                 gamma=None,
                 vf_loss_coeff=None,
                 clip_rho_threshold=None,
                 clip_pg_rho_threshold=None):
        # same argument checks as impala.py:100-104
        assert isinstance(sample_batch_steps, int)
        assert isinstance(gamma, float)
        assert isinstance(vf_loss_coeff, float)
        assert isinstance(clip_rho_threshold, float)
        assert isinstance(clip_pg_rho_threshold, float)
        super(IMPALA, self).__init__(model)
        self.sample_batch_steps = sample_batch_steps
        self.gamma = gamma
        self.vf_loss_coeff = vf_loss_coeff
        self.clip_rho_threshold = clip_rho_threshold
        self.clip_pg_rho_threshold = clip_pg_rho_threshold
        # paddle Adam(lr=0.001, ClipGradByGlobalNorm(40)) (impala.py:113-117); eps defaults agree
        self.optimizer = torch.optim.Adam(self.model.parameters(), lr=0.001)
        self.grad_clip_norm = 40.0
        self.grad_hook = None  # set by parl_amd.dist for data-parallel learners
        # one-kernel loss + gradient (ops.impala_loss); False keeps the autograd graph of the
        # reference formulas on top of the fused V-trace (the parity tests compare the two)
        self.fused_loss = True
        # rows per forward / backward pass of learn() (None: the whole batch at once).  Models whose
        # convolutions are GEMM-lowered (the 84x84 network) need ~2 MB of im2col per row.
        self.max_learn_rows = None
        # how learn() uses max_learn_rows.  'forward': the network runs chunk by chunk (separate autograd
        # nodes, so the backward kernels also run chunk by chunk), the heads' outputs are concatenated and
        # the fused V-trace loss runs ONCE on the whole [T, B] batch — activations of the whole batch are
        # kept (15.5 KB per row for the 42x42 model).  'accumulate': forward + loss + backward per chunk
        # (bounded memory: what the GEMM-lowered float path needs, ~2 MB of im2col per row).
        self.learn_chunk_mode = 'forward'
        # heads + loss + heads' backward in one kernel when the model exposes its trunk (`_trunk`,
        # `policy_fc`, `value_fc`), the batch is time-major, T <= 64, 256 hidden units, A in {4, 6}
        self.fused_heads = True

    def _heads(self, obs):
        if hasattr(self.model, 'policy_and_value'):
            return self.model.policy_and_value(obs)
        return self.model.policy(obs), self.model.value(obs)

    def _heads_in_chunks(self, obs, time_major, trunk_only=False):
        """policy logits [N, A] and values [N] of a flat batch (trunk_only: the trunk's output [N, H]); with
        max_learn_rows (mode 'forward') the network sees chunks of whole sequences, the outputs come
        back in the batch's row order"""
        T = self.sample_batch_steps
        N = obs.shape[0]
        B = N // T
        per = max(1, (self.max_learn_rows or N) // T)
        if self.learn_chunk_mode != 'forward' or per >= B:
            return self.model._trunk(obs) if trunk_only else self._heads(obs)
        ls, vs = [], []
        if time_major:
            o = obs.reshape((T, B) + tuple(obs.shape[1:]))
            for b0 in range(0, B, per):
                b1 = min(B, b0 + per)
                oc = o[:, b0:b1].reshape((T * (b1 - b0), ) + tuple(obs.shape[1:]))
                if trunk_only:
                    ls.append(self.model._trunk(oc).reshape(T, b1 - b0, -1))
                    continue
                l, v = self._heads(oc)
                ls.append(l.reshape(T, b1 - b0, -1))
                vs.append(v.reshape(T, b1 - b0))
            if trunk_only:
                return torch.cat(ls, 1).reshape(N, -1)
            return torch.cat(ls, 1).reshape(N, -1), torch.cat(vs, 1).reshape(N)
        assert not trunk_only
        for b0 in range(0, B, per):
            l, v = self._heads(obs[b0 * T:min(B, b0 + per) * T])
            ls.append(l)
            vs.append(v)
        return torch.cat(ls, 0), torch.cat(vs, 0)

    def _can_fuse_heads(self, obs, time_major):
        """the one-kernel heads + loss path (ops.impala_heads_loss) applies: time-major batch on the device, T <= 64,
        a model that exposes its trunk and two biased float32 Linear heads on 256 hidden units, A in {4, 6}"""
        m, T = self.model, self.sample_batch_steps
        pf, vf = getattr(m, 'policy_fc', None), getattr(m, 'value_fc', None)
        return bool(self.fused_loss and self.fused_heads and time_major and T <= 64 and obs.is_cuda
                    and hasattr(m, '_trunk') and isinstance(pf, torch.nn.Linear) and isinstance(vf, torch.nn.Linear)
                    and pf.in_features == 256 and vf.in_features == 256 and pf.out_features in (4, 6)
                    and vf.out_features == 1 and pf.bias is not None and vf.bias is not None
                    and pf.weight.dtype == torch.float32 and vf.weight.dtype == torch.float32
                    and not torch.is_autocast_enabled())

    def _vtrace_loss(self, obs, actions, behaviour_logits, rewards, dones, entropy_coeff, time_major):
        """forward pass + fused V-trace + loss terms for one flat batch (no parameter update)"""
        T = self.sample_batch_steps
        N = obs.shape[0]
        B = N // T
        m = self.model
        hidden = None
        if self._can_fuse_heads(obs, time_major):
            hidden = self._heads_in_chunks(obs, True, trunk_only=True)
            if hidden.dtype == torch.float32:
                A = m.policy_fc.out_features
                try:
                    total, sums, vs, pg_adv = _FusedHeadsLossFn.apply(
                        hidden.reshape(T, B, 256), m.policy_fc.weight, m.policy_fc.bias, m.value_fc.weight,
                        m.value_fc.bias, behaviour_logits.reshape(T, B, A), actions.reshape(T, B), rewards.reshape(T, B),
                        dones.reshape(T, B), (self.gamma, self.clip_rho_threshold, self.clip_pg_rho_threshold,
                                              self.vf_loss_coeff, float(entropy_coeff)))
                    return _KernelVTraceLoss(total, sums, vs, pg_adv), (sums[3] / N).float()
                except NotImplementedError:  # no instantiation for this shape: heads by the framework, below
                    pass
        if hidden is not None:  # the fused path declined: the framework's heads on the trunk output we already have
            target_logits, values = m.policy_fc(hidden), m.value_fc(hidden).squeeze(1)
        else:
            target_logits, values = self._heads_in_chunks(obs, time_major)
        A = target_logits.shape[-1]
        if self.fused_loss and T <= 256 and A in (2, 3, 4, 6, 9, 18):
            # log-softmax gather, entropy, KL, V-trace, the three sums AND their gradient: one kernel
            shp = (T, B) if time_major else (B, T)
            total, sums, vs, pg_adv = _FusedLossFn.apply(
                target_logits.reshape(shp + (A, )), values.reshape(shp), behaviour_logits.reshape(shp + (A, )),
                actions.reshape(shp), rewards.reshape(shp), dones.reshape(shp),
                (self.gamma, self.clip_rho_threshold, self.clip_pg_rho_threshold, self.vf_loss_coeff,
                 float(entropy_coeff), time_major))
            return _KernelVTraceLoss(total, sums, vs, pg_adv), (sums[3] / N).float()
        logp_all = F.log_softmax(target_logits, dim=-1)
        p_all = logp_all.exp()
        policy_entropy = -(p_all * logp_all).sum(-1)  # Categorical.entropy (impala.py:156)
        target_actions_log_probs = logp_all.gather(-1, actions.unsqueeze(-1)).squeeze(-1)  # :119-132
        with torch.no_grad():  # kl for debug (impala.py:161-165)
            blogp = F.log_softmax(behaviour_logits, dim=-1)
            kl = (p_all * (logp_all - blogp)).sum(-1).mean()
        if time_major:
            shp, cut = (T, B), (lambda t: t.reshape(T, B)[:-1])
        else:
            shp, cut = (B, T), (lambda t: t.reshape(B, T)[:, :-1])
        with torch.no_grad():
            vs, pg_adv = ops.vtrace_from_logits(
                behaviour_logits.reshape(shp + (A, )), target_logits.detach().reshape(shp + (A, )),
                actions.reshape(shp), rewards.reshape(shp), dones.reshape(shp), values.detach().reshape(shp),
                self.gamma, self.clip_rho_threshold, self.clip_pg_rho_threshold, time_major=time_major)
        # drop the last step of every sequence: it only supplies the bootstrap (impala.py:186-194)
        vtrace_loss = _FusedVTraceLoss(vs, pg_adv, cut(target_actions_log_probs), cut(values), cut(policy_entropy),
                                       entropy_coeff, self.vf_loss_coeff)
        return vtrace_loss, kl

    def _apply_gradients(self, learning_rate):
        if self.grad_hook is not None:
            self.grad_hook(self.model)  # e.g. RCCL all-reduce (sum) of the flattened gradient
        from .graphed import set_lr
        set_lr(self.optimizer, learning_rate)  # a device scalar once a GraphedLearn made the optimizer capturable
        torch.nn.utils.clip_grad_norm_(self.model.parameters(), max_norm=self.grad_clip_norm)
        self.optimizer.step()

    def learn(self, obs, actions, behaviour_logits, rewards, dones, learning_rate, entropy_coeff,
              time_major=False):
        """Reference contract (impala.py:134-215): flat batches of N = B*T rows, ENV-major
        ([env0 t0..tT-1, env1 ...], examples/IMPALA/actor.py:78-89).  time_major=True takes the
        device rollout layout instead (rows ordered [t0 all envs, t1 all envs, ...]).

        obs [N,C,H,W] (uint8 or float32), actions int64 [N], behaviour_logits f32 [N,A],
        rewards f32 [N], dones bool [N].  Returns (vtrace_loss, kl)."""
        T = self.sample_batch_steps
        N = obs.shape[0]
        if self.learn_chunk_mode == 'accumulate' and self.max_learn_rows and N > self.max_learn_rows and N // T > 1:
            # bounded-memory update: the batch in chunks of whole sequences, gradients accumulated
            # (the losses are sums over rows, impala.py:67-79), ONE clip + Adam step
            B = N // T
            per = max(1, self.max_learn_rows // T)
            chunks = []
            for b0 in range(0, B, per):
                b1 = min(B, b0 + per)
                if time_major:
                    cut = lambda t: t.reshape((T, B) + tuple(t.shape[1:]))[:, b0:b1].reshape((T * (b1 - b0), ) + tuple(t.shape[1:]))  # noqa: E731
                else:
                    cut = lambda t: t[b0 * T:b1 * T]  # noqa: E731
                chunks.append({'obs': cut(obs), 'actions': cut(actions), 'behaviour_logits': cut(behaviour_logits),
                               'rewards': cut(rewards), 'dones': cut(dones)})
            return self.learn_batches(chunks, learning_rate, entropy_coeff, time_major=time_major)
        vtrace_loss, kl = self._vtrace_loss(obs, actions, behaviour_logits, rewards, dones, entropy_coeff, time_major)
        self._zero_grad()
        vtrace_loss.total_loss.backward()
        self._apply_gradients(learning_rate)
        # what the caller gets are VALUES (the reference's agent does `.cpu().numpy()` on them,
        # examples/IMPALA/atari_agent.py:69-73): nothing may still hang on the autograd graph
        for k in ('total_loss', 'pi_loss', 'vf_loss', 'entropy'):
            setattr(vtrace_loss, k, getattr(vtrace_loss, k).detach())
        return vtrace_loss, kl.detach()

    def learn_batches(self, batches, learning_rate, entropy_coeff, time_major=False):
        """ONE parameter update on the union of several flat batches (dicts with the arguments of
        learn()).  The losses are sums over rows (impala.py:67-79), so the gradient of the union is
        the sum of the per-batch gradients: each batch is forwarded / backpropagated on its own
        (accumulating into .grad), then clip + Adam run once.  Used by the multi-group actor
        pipeline, whose groups deliver their trajectories in separate buffers."""
        self._zero_grad()
        out, kls = None, []
        for b in batches:
            loss, kl = self._vtrace_loss(b['obs'], b['actions'], b['behaviour_logits'], b['rewards'], b['dones'],
                                         entropy_coeff, time_major)
            loss.total_loss.backward()
            kls.append(kl)
            if out is None:
                out = loss
            else:  # report the sums over the union, like one big batch would
                for k in ('total_loss', 'pi_loss', 'vf_loss', 'entropy'):
                    setattr(out, k, getattr(out, k).detach() + getattr(loss, k).detach())
        self._apply_gradients(learning_rate)
        return out, torch.stack(kls).mean()

    @torch.no_grad()
    def sample(self, obs):
        """(probs, logits) — impala.py:217-227"""
        logits = self.model.policy(obs)
        return F.softmax(logits, dim=-1), logits

    @torch.no_grad()
    def predict(self, obs):
        """greedy action (the fluid IMPALA exposes predict, fluid impala.py:214-225)"""
        return self.model.policy(obs).argmax(-1)
