"""Tensor-level wrappers over the C ABI (include/parl_hip.h).

Every function takes HIP-resident torch tensors, enqueues the hand-written gfx950 kernel on
torch's current stream and returns torch tensors.  Nothing here computes on the host and
nothing falls back: a CPU tensor or a missing library raises ``ParlHipError``/``ImportError``.
"""

import os

import torch

from . import _native as N

_NAN = float('nan')


def _f32(t, name):
    if t.dtype != torch.float32:
        raise N.ParlHipError('%s must be float32, got %s' % (name, t.dtype))
    return t.contiguous()


def _thr(x):
    """Reference `None` threshold (vtrace.py:102-105) -> NaN sentinel of the C ABI."""
    return _NAN if x is None else float(x)


def vtrace(behaviour_actions_log_probs, target_actions_log_probs, discounts, rewards,
           values, bootstrap_value, clip_rho_threshold=1.0, clip_pg_rho_threshold=1.0):
    """vtrace.from_importance_weights (parl/algorithms/paddle/impala/vtrace.py:36-139).

    All [T,B] float32 time-major; returns (vs, pg_advantages)."""
    blp = _f32(behaviour_actions_log_probs, 'behaviour_actions_log_probs')
    tlp = _f32(target_actions_log_probs, 'target_actions_log_probs')
    disc = _f32(discounts, 'discounts')
    rew = _f32(rewards, 'rewards')
    val = _f32(values, 'values')
    boot = _f32(bootstrap_value, 'bootstrap_value')
    T, B = blp.shape
    vs = torch.empty_like(val)
    pg = torch.empty_like(val)
    N.check(
        N.lib().parlhip_vtrace_f32(
            N.ptr(blp), N.ptr(tlp), N.ptr(disc), N.ptr(rew), N.ptr(val), N.ptr(boot),
            N.ptr(vs), N.ptr(pg), T, B, _thr(clip_rho_threshold),
            _thr(clip_pg_rho_threshold), N.stream_ptr()), 'parlhip_vtrace_f32')
    return vs, pg


def vtrace_from_logits(behaviour_logits, target_logits, actions, rewards, dones, values,
                       gamma, clip_rho_threshold=1.0, clip_pg_rho_threshold=1.0,
                       time_major=True, want_log_probs=False):
    """Fused _log_prob ×2 + discounts + drop-last/bootstrap + V-trace
    (impala.py:59,119-132,167-194 + vtrace.py:36-139).

    Shapes: logits [T,B,A] (time_major) or [B,T,A]; actions int64, rewards f32, dones
    bool/uint8, values f32 all [T,B] / [B,T].  Returns vs, pg_advantages (and the two
    log-prob tensors if requested) with T-1 steps in the same major order."""
    bl = _f32(behaviour_logits, 'behaviour_logits')
    tl = _f32(target_logits, 'target_logits')
    if time_major:
        T, B, A = tl.shape
        oshape = (T - 1, B)
    else:
        B, T, A = tl.shape
        oshape = (B, T - 1)
    if actions.dtype != torch.int64:
        raise N.ParlHipError('actions must be int64')
    actions = actions.contiguous()
    rew = _f32(rewards, 'rewards')
    val = _f32(values, 'values')
    if dones.dtype == torch.bool:
        dones = dones.contiguous().view(torch.uint8)
    elif dones.dtype != torch.uint8:
        raise N.ParlHipError('dones must be bool or uint8')
    dones = dones.contiguous()
    vs = torch.empty(oshape, dtype=torch.float32, device=val.device)
    pg = torch.empty(oshape, dtype=torch.float32, device=val.device)
    tlp = torch.empty(oshape, dtype=torch.float32, device=val.device) if want_log_probs else None
    blp = torch.empty(oshape, dtype=torch.float32, device=val.device) if want_log_probs else None
    N.check(
        N.lib().parlhip_vtrace_from_logits_f32(
            N.ptr(bl), N.ptr(tl), N.ptr(actions), N.ptr(rew), N.ptr(dones), N.ptr(val),
            N.ptr(vs), N.ptr(pg), N.ptr(tlp), N.ptr(blp), T, B, A, 1 if time_major else 0,
            float(gamma), _thr(clip_rho_threshold), _thr(clip_pg_rho_threshold),
            N.stream_ptr()), 'parlhip_vtrace_from_logits_f32')
    if want_log_probs:
        return vs, pg, tlp, blp
    return vs, pg


ENOSUP = -3  # PARLHIP_ENOSUP (include/parl_hip.h)


def impala_loss(behaviour_logits, target_logits, actions, rewards, dones, values, gamma, clip_rho_threshold=1.0,
                clip_pg_rho_threshold=1.0, vf_coeff=0.5, entropy_coeff=-0.01, time_major=True):
    """VTraceLoss + the pre-processing of IMPALA.learn (impala.py:25-79,119-194) in one launch,
    including the gradient of total_loss w.r.t. target_logits and values.  Shapes as
    vtrace_from_logits.  Returns (vs, pg_advantages, grad_logits, grad_values, sums) with sums =
    f64[4] device tensor (pi_loss, vf_loss, entropy, KL summed over all rows), or None when the
    library has no instantiation for this (T, A) (callers use the unfused path)."""
    bl, tl = _f32(behaviour_logits, 'behaviour_logits'), _f32(target_logits, 'target_logits')
    if time_major:
        T, B, A = tl.shape
        oshape = (T - 1, B)
    else:
        B, T, A = tl.shape
        oshape = (B, T - 1)
    if actions.dtype != torch.int64:
        raise N.ParlHipError('actions must be int64')
    actions = actions.contiguous()
    rew, val = _f32(rewards, 'rewards'), _f32(values, 'values')
    if dones.dtype == torch.bool:
        dones = dones.contiguous().view(torch.uint8)
    elif dones.dtype != torch.uint8:
        raise N.ParlHipError('dones must be bool or uint8')
    dones = dones.contiguous()
    dev = val.device
    vs = torch.empty(oshape, dtype=torch.float32, device=dev)
    pg = torch.empty(oshape, dtype=torch.float32, device=dev)
    glog = torch.empty_like(tl)
    gval = torch.empty_like(val)
    sums = torch.zeros(4, dtype=torch.float64, device=dev)
    code = N.lib().parlhip_impala_loss_f32(
        N.ptr(bl), N.ptr(tl), N.ptr(actions), N.ptr(rew), N.ptr(dones), N.ptr(val), N.ptr(vs), N.ptr(pg), N.ptr(glog),
        N.ptr(gval), N.ptr(sums), T, B, A, 1 if time_major else 0, float(gamma), _thr(clip_rho_threshold),
        _thr(clip_pg_rho_threshold), float(vf_coeff), float(entropy_coeff), N.stream_ptr())
    if code == ENOSUP:
        return None
    N.check(code, 'parlhip_impala_loss_f32')
    return vs, pg, glog, gval, sums


def impala_heads_loss(hidden, w_policy, b_policy, w_value, b_value, behaviour_logits, actions, rewards, dones, gamma,
                      clip_rho_threshold=1.0, clip_pg_rho_threshold=1.0, vf_coeff=0.5, entropy_coeff=-0.01):
    """policy_fc + value_fc + impala_loss + the heads' backward in one launch (time-major).  hidden f32
    [T,B,256]; w_policy [A,256], b_policy [A], w_value [1,256] or [256], b_value [1]; the rest as
    impala_loss with shapes [T,B(,A)].  Returns (vs, pg_advantages, grad_hidden, grad_w_policy,
    grad_b_policy, grad_w_value, grad_b_value, sums) or None when the library has no instantiation."""
    hd = _f32(hidden, 'hidden')
    T, B, H = hd.shape
    wp, bp = _f32(w_policy, 'w_policy'), _f32(b_policy, 'b_policy')
    wv, bv = _f32(w_value, 'w_value').reshape(-1), _f32(b_value, 'b_value').reshape(-1)
    A = wp.shape[0]
    if wp.shape != (A, H) or wv.numel() != H or bp.numel() != A or bv.numel() != 1:
        raise N.ParlHipError('impala_heads_loss: head shapes %r %r %r %r for hidden %r' %
                             (tuple(wp.shape), tuple(bp.shape), tuple(wv.shape), tuple(bv.shape), tuple(hd.shape)))
    bl = _f32(behaviour_logits, 'behaviour_logits')
    if tuple(bl.shape) != (T, B, A):
        raise N.ParlHipError('behaviour_logits must be [T,B,A]')
    if actions.dtype != torch.int64:
        raise N.ParlHipError('actions must be int64')
    actions = actions.contiguous()
    rew = _f32(rewards, 'rewards')
    if dones.dtype == torch.bool:
        dones = dones.contiguous().view(torch.uint8)
    elif dones.dtype != torch.uint8:
        raise N.ParlHipError('dones must be bool or uint8')
    dones = dones.contiguous()
    dev = hd.device
    L = N.lib()
    nws = L.parlhip_impala_heads_loss_workspace_bytes(B, A)
    if nws == 0:
        return None
    vs = torch.empty((T - 1, B), dtype=torch.float32, device=dev)
    pg = torch.empty((T - 1, B), dtype=torch.float32, device=dev)
    gh = torch.empty_like(hd)
    gheads = torch.empty((A + 1) * H + (A + 1), dtype=torch.float32, device=dev)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    sums = torch.zeros(4, dtype=torch.float64, device=dev)
    code = L.parlhip_impala_heads_loss_f32(
        N.ptr(hd), N.ptr(wp), N.ptr(bp), N.ptr(wv), N.ptr(bv), N.ptr(bl), N.ptr(actions), N.ptr(rew), N.ptr(dones),
        N.ptr(vs), N.ptr(pg), N.ptr(gh), N.ptr(gheads), N.ptr(sums), N.ptr(ws), T, B, H, A, float(gamma),
        _thr(clip_rho_threshold), _thr(clip_pg_rho_threshold), float(vf_coeff), float(entropy_coeff), N.stream_ptr())
    if code == ENOSUP:
        return None
    N.check(code, 'parlhip_impala_heads_loss_f32')
    gw = gheads[:(A + 1) * H].view(A + 1, H)
    gb = gheads[(A + 1) * H:]
    return vs, pg, gh, gw[:A], gb[:A], gw[A:], gb[A:], sums


GAE_DONE_ENDS_STEP = 0
GAE_DONE_STARTS_STEP = 1


def gae(rewards, values, dones, next_value, gamma, lam, last_done=None,
        done_convention=GAE_DONE_ENDS_STEP, allow_chunked=True):
    """Batched calc_gae (rl_utils.py:34-51 with examples/A2C/actor.py:73-85 segments) or
    RolloutStorage.compute_returns (examples/PPO/storage.py:45-64).

    rewards/values [T,B] f32, dones [T,B] bool/uint8/float32, next_value [B].
    Returns (advantages, returns) with returns = advantages + values.
    allow_chunked=False forces the single-pass kernel (bit-exact fp32 op order of numpy in the
    PPO convention); the chunk-parallel plan is chosen automatically for long T over few sequences."""
    rew = _f32(rewards, 'rewards')
    val = _f32(values, 'values')
    nv = _f32(next_value, 'next_value').reshape(-1)
    T, B = rew.shape
    if dones.dtype == torch.bool:
        dones = dones.contiguous().view(torch.uint8)
    dones = dones.contiguous()
    is_f32 = dones.dtype == torch.float32
    if not is_f32 and dones.dtype != torch.uint8:
        raise N.ParlHipError('dones must be bool, uint8 or float32')
    if last_done is not None:
        if last_done.dtype == torch.bool:
            last_done = last_done.contiguous().view(torch.uint8)
        last_done = last_done.contiguous().reshape(-1)
        if last_done.dtype != dones.dtype:
            raise N.ParlHipError('last_done must have the dtype of dones')
    adv = torch.empty_like(val)
    ret = torch.empty_like(val)
    wsb = N.lib().parlhip_gae_workspace_bytes(T, B) if allow_chunked else 0
    if wsb:  # long T over few sequences (PPO storage shape): chunk-parallel plan
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=val.device)
        N.check(
            N.lib().parlhip_gae_ws_f32(
                N.ptr(rew), N.ptr(val), N.ptr(dones), N.ptr(nv), N.ptr(last_done),
                N.ptr(adv), N.ptr(ret), T, B, float(gamma), float(lam), int(done_convention),
                1 if is_f32 else 0, N.ptr(ws), wsb, N.stream_ptr()), 'parlhip_gae_ws_f32')
        return adv, ret
    N.check(
        N.lib().parlhip_gae_f32(
            N.ptr(rew), N.ptr(val), N.ptr(dones), N.ptr(nv), N.ptr(last_done),
            N.ptr(adv), N.ptr(ret), T, B, float(gamma), float(lam), int(done_convention),
            1 if is_f32 else 0, N.stream_ptr()), 'parlhip_gae_f32')
    return adv, ret


def discount_cumsum(x, gamma, dones=None):
    """Batched calc_discount_sum_rewards (rl_utils.py:21-31) over [T,B]."""
    x = _f32(x, 'x')
    T, B = x.shape
    if dones is not None:
        if dones.dtype == torch.bool:
            dones = dones.contiguous().view(torch.uint8)
        dones = dones.contiguous()
    out = torch.empty_like(x)
    N.check(
        N.lib().parlhip_discount_cumsum_f32(
            N.ptr(x), N.ptr(dones), N.ptr(out), T, B, float(gamma), N.stream_ptr()),
        'parlhip_discount_cumsum_f32')
    return out


def adv_normalize(adv, idx=None, eps=1e-8, return_stats=False):
    """(adv - mean) / (std + eps) with the unbiased std (ppo.py:124-127), optionally on the
    gathered minibatch adv[idx]."""
    adv = _f32(adv, 'adv').reshape(-1)
    if idx is not None:
        if idx.dtype != torch.int64:
            raise N.ParlHipError('idx must be int64')
        idx = idx.contiguous().reshape(-1)
        n = idx.numel()
    else:
        n = adv.numel()
    out = torch.empty(n, dtype=torch.float32, device=adv.device)
    wsb = N.lib().parlhip_adv_normalize_workspace_bytes(n)
    ws = torch.empty(max(wsb // 8, 1), dtype=torch.float64, device=adv.device)
    stats = torch.empty(2, dtype=torch.float32, device=adv.device) if return_stats else None
    N.check(
        N.lib().parlhip_adv_normalize_f32(
            N.ptr(adv), N.ptr(idx), N.ptr(out), n, float(eps), N.ptr(ws), wsb,
            N.ptr(stats), N.stream_ptr()), 'parlhip_adv_normalize_f32')
    if return_stats:
        return out, stats
    return out


class ClipAdam(object):
    """torch.nn.utils.clip_grad_norm_(params, max_norm) + torch.optim.Adam.step() as TWO launches
    (parlhip_clip_adam_f32; the pair parl/algorithms/paddle/impala.py:113-117 configures) over the optimizer's OWN
    state tensors — `exp_avg`, `exp_avg_sq`, the float32 device `step` scalars and the device `lr` scalar of a
    capturable torch Adam (graphed.make_capturable): state_dict(), checkpoints and load_optimizer_state_inplace see
    nothing new.  Missing state is created as torch's first step would (zeros).  `ClipAdam.supported(optimizer)`
    says whether the optimizer is one this covers (one param group, <= 16 float32 parameters, no weight decay /
    amsgrad / maximize); callers keep the framework pair otherwise."""

    MAX_TENSORS = 16

    @staticmethod
    def supported(optimizer):
        if not isinstance(optimizer, torch.optim.Adam) or type(optimizer) is not torch.optim.Adam:
            return False
        if len(optimizer.param_groups) != 1:
            return False
        g = optimizer.param_groups[0]
        ps = [p for p in g['params'] if p.requires_grad]
        if not ps or len(ps) > ClipAdam.MAX_TENSORS:
            return False
        if g.get('weight_decay', 0) or g.get('amsgrad', False) or g.get('maximize', False):
            return False
        if g.get('decoupled_weight_decay', False) or g.get('differentiable', False):
            return False
        if not isinstance(g['lr'], torch.Tensor) or not g['lr'].is_cuda or g['lr'].dtype != torch.float32:
            return False
        return all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in ps)

    def __init__(self, optimizer, max_norm):
        import ctypes
        assert ClipAdam.supported(optimizer)
        self.optimizer, self.max_norm = optimizer, float(max_norm)
        g = optimizer.param_groups[0]
        self.params = [p for p in g['params'] if p.requires_grad]
        dev = self.params[0].device
        for p in self.params:  # torch.optim.Adam._init_group for a capturable optimizer
            st = optimizer.state[p]
            if len(st) == 0:
                st['step'] = torch.zeros((), dtype=torch.float32, device=dev)
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if not (st['step'].is_cuda and st['step'].dtype == torch.float32):
                raise N.ParlHipError('ClipAdam: the optimizer is not capturable (graphed.make_capturable first)')
        n = len(self.params)
        self._numel = (ctypes.c_int64 * n)(*[p.numel() for p in self.params])
        wsb = N.lib().parlhip_clip_adam_workspace_bytes(n, ctypes.cast(self._numel, ctypes.c_void_p))
        self.workspace = torch.empty(max(wsb // 4, 1), dtype=torch.float32, device=dev)
        self.norm = torch.zeros(1, dtype=torch.float32, device=dev)   # the global norm before clipping, last step
        self._ct = ctypes
        self._tables = None
        for p in self.params:   # step() writes the parameters through raw pointers: their autograd version never moves,
            p._parl_graph_written = True   # so a cached MFMA layout of them (_cached_layout) would go stale — never cache

    def _pointer_tables(self):
        """HOST arrays of device pointers, rebuilt when a tensor moved (p.grad is re-created by zero_grad(set_to_none))"""
        ct, opt = self._ct, self.optimizer
        for p in self.params:
            if p.grad is None or not p.grad.is_contiguous() or p.grad.dtype != torch.float32:
                raise N.ParlHipError('ClipAdam.step: every parameter needs a contiguous float32 .grad')
        cols = ([p.data_ptr() for p in self.params], [p.grad.data_ptr() for p in self.params],
                [opt.state[p]['exp_avg'].data_ptr() for p in self.params],
                [opt.state[p]['exp_avg_sq'].data_ptr() for p in self.params],
                [opt.state[p]['step'].data_ptr() for p in self.params])
        key = tuple(tuple(c) for c in cols)   # (a plain optimizer.load_state_dict before a capture replaces state tensors)
        if self._tables is None or self._tables[0] != key:
            n = len(self.params)
            self._tables = (key, [(ct.c_void_p * n)(*c) for c in cols])
        return self._tables[1]

    @torch.no_grad()
    def step(self):
        ct = self._ct
        g = self.optimizer.param_groups[0]
        b1, b2 = g['betas']
        tp, tg, tm, tv, ts = self._pointer_tables()
        cast = lambda a: ct.cast(a, ct.c_void_p)  # noqa: E731
        N.check(N.lib().parlhip_clip_adam_f32(len(self.params), cast(tp), cast(tg), cast(tm), cast(tv), cast(ts),
                                              cast(self._numel), N.ptr(g['lr']), float(b1), float(b2), float(g['eps']),
                                              self.max_norm, N.ptr(self.workspace), N.ptr(self.norm), N.stream_ptr()),
                'parlhip_clip_adam_f32')


def categorical_sample(probs, uniforms):
    """np.random.choice(A, 1, p=prob) per row given explicit float64 uniforms
    (examples/IMPALA/atari_agent.py:38-40).  Returns int64 [B]."""
    probs = _f32(probs, 'probs')
    B, A = probs.shape
    if uniforms.dtype != torch.float64:
        raise N.ParlHipError('uniforms must be float64')
    uniforms = uniforms.contiguous()
    actions = torch.empty(B, dtype=torch.int64, device=probs.device)
    N.check(
        N.lib().parlhip_categorical_sample_f32(
            N.ptr(probs), N.ptr(uniforms), N.ptr(actions), B, A, N.stream_ptr()),
        'parlhip_categorical_sample_f32')
    return actions


def policy_sample(logits_or_probs, seed, offset, row0=0, is_logits=True, want_probs=False,
                  want_uniforms=False):
    """IMPALA.sample's softmax + per-row np.random.choice with on-device Philox uniforms.

    u[b] = philox4x32-10(seed; offset, row0+b) -> 53-bit double.  Returns actions (and the
    probabilities / uniforms when requested, for parity checks)."""
    x = _f32(logits_or_probs, 'logits_or_probs')
    B, A = x.shape
    actions = torch.empty(B, dtype=torch.int64, device=x.device)
    probs = torch.empty_like(x) if want_probs else None
    uni = torch.empty(B, dtype=torch.float64, device=x.device) if want_uniforms else None
    N.check(
        N.lib().parlhip_policy_sample_f32(
            N.ptr(x), 1 if is_logits else 0, N.ptr(actions), N.ptr(probs), N.ptr(uni), B, A,
            int(seed) & (2**64 - 1), int(offset) & (2**64 - 1), int(row0) & (2**64 - 1),
            N.stream_ptr()), 'parlhip_policy_sample_f32')
    out = (actions, )
    if want_probs:
        out += (probs, )
    if want_uniforms:
        out += (uni, )
    return out if len(out) > 1 else actions


def policy_sample_into(logits, actions_out, seed, offset, row0=0, offset_base=None):
    """policy_sample writing into a preallocated int64 [B] slab (rollout buffers).  offset_base: an int64 [1] device
    tensor added to `offset` on the device (a rollout replayed as a hipGraph keeps the number of its first step there)."""
    x = _f32(logits, 'logits')
    B, A = x.shape
    if offset_base is not None:
        if offset_base.dtype != torch.int64 or offset_base.numel() != 1 or not offset_base.is_cuda:
            raise N.ParlHipError('offset_base must be an int64 [1] device tensor')
        N.check(
            N.lib().parlhip_policy_sample_at_f32(
                N.ptr(x), 1, N.ptr(actions_out), None, None, B, A, int(seed) & (2**64 - 1), N.ptr(offset_base),
                int(offset) & (2**64 - 1), int(row0) & (2**64 - 1), N.stream_ptr()), 'parlhip_policy_sample_at_f32')
        return actions_out
    N.check(
        N.lib().parlhip_policy_sample_f32(
            N.ptr(x), 1, N.ptr(actions_out), None, None, B, A, int(seed) & (2**64 - 1),
            int(offset) & (2**64 - 1), int(row0) & (2**64 - 1), N.stream_ptr()), 'parlhip_policy_sample_f32')
    return actions_out


def policy_head_sample_into(hidden, w_policy, b_policy, logits_out, actions_out, seed, offset, row0=0, offset_base=None):
    """policy_fc + policy_sample in one launch for the actors: logits_out [B,A] f32 and actions_out [B] int64 are
    slabs of a rollout buffer; hidden f32 [B,256].  offset_base: an int64 [1] device tensor added to `offset` on the
    device (a rollout replayed as a hipGraph keeps the number of its first step there).  Returns False when the
    library has no instantiation (the caller then runs the framework's head + policy_sample_into)."""
    hd = _f32(hidden, 'hidden')
    B, H = hd.shape
    A = w_policy.shape[0]
    if H != 256 or A > 18 or logits_out.dtype != torch.float32 or not logits_out.is_contiguous():
        return False
    wp, bp = _f32(w_policy.detach(), 'w_policy'), _f32(b_policy.detach(), 'b_policy')
    if offset_base is not None:
        if offset_base.dtype != torch.int64 or offset_base.numel() != 1 or not offset_base.is_cuda:
            raise N.ParlHipError('offset_base must be an int64 [1] device tensor')
        code = N.lib().parlhip_policy_head_sample_at_f32(N.ptr(hd), N.ptr(wp), N.ptr(bp), N.ptr(logits_out),
                                                         N.ptr(actions_out), B, H, A, int(seed) & (2**64 - 1),
                                                         N.ptr(offset_base), int(offset) & (2**64 - 1),
                                                         int(row0) & (2**64 - 1), N.stream_ptr())
    else:
        code = N.lib().parlhip_policy_head_sample_f32(N.ptr(hd), N.ptr(wp), N.ptr(bp), N.ptr(logits_out), N.ptr(actions_out),
                                                      B, H, A, int(seed) & (2**64 - 1), int(offset) & (2**64 - 1),
                                                      int(row0) & (2**64 - 1), N.stream_ptr())
    if code == ENOSUP:
        return False
    N.check(code, 'parlhip_policy_head_sample_f32')
    return True


class RingObservation(object):
    """The CURRENT stacked observation of every env of a rollout ring, not materialised: ring u8 [S, E, d*d] single
    frames, since u8 [S, E] (steps since the env's last reset, clamped at 3), `slot` the ring position of the newest
    frame — frame j of env n's stack is the frame min(3 - j, since[slot, n]) slots back (FrameStack,
    parl/env/atari_wrappers.py, as DeviceVectorEnv keeps it).  atari42_conv12 reads it in place (one launch less per
    env step); `materialize()` is the gather every other consumer gets."""

    def __init__(self, ring, since, slot, dim, gather):
        self.ring, self.since, self.slot, self.dim = ring, since, int(slot), int(dim)
        self._gather = gather
        self.shape = (ring.shape[1], 4, self.dim, self.dim)
        self.dtype, self.device, self.is_cuda = torch.uint8, ring.device, ring.is_cuda

    def materialize(self, out=None):
        return self._gather(out)


def atari42_conv12_pack(conv1_weight, conv2_weight, out=None):
    """The two weight matrices of atari42_conv12 in the kernel's operand order (parlhip_atari42_conv12_weights_f32):
    flat f32 [17,408] — the forward kernel's [36][64] float4s, then the backward kernel's [4][8][64].  A workgroup then
    fetches its operands with 36 (backward: 12) coalesced loads instead of 144 (48) scattered ones — half of the
    actors' 1024-observation launch was that fetch.  Rebuild when the weights change."""
    if tuple(conv1_weight.shape) != (16, 4, 4, 4) or tuple(conv2_weight.shape) != (32, 16, 4, 4):
        raise N.ParlHipError('atari42_conv12_pack: weights must be [16,4,4,4] and [32,16,4,4]')
    w1, w2 = _f32(conv1_weight.detach(), 'conv1_weight'), _f32(conv2_weight.detach(), 'conv2_weight')
    n = N.lib().parlhip_atari42_conv12_weights_bytes() // 4
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=w1.device)
    elif out.dtype != torch.float32 or out.numel() != n or not out.is_contiguous():
        raise N.ParlHipError('atari42_conv12_pack: out must be contiguous f32 [%d]' % n)
    N.check(N.lib().parlhip_atari42_conv12_weights_f32(N.ptr(w1), N.ptr(w2), N.ptr(out), N.stream_ptr()),
            'parlhip_atari42_conv12_weights_f32')
    return out


A1_SAVE_MAX_ROWS = 8192   # rows up to which the learner's forward keeps the conv1 activation (40 KB per row) for its backward


def atari42_conv12(obs, conv1_weight, conv1_bias, conv2_weight, conv2_bias, out=None, packed=None, save_a1=False):
    """conv1 + ReLU + conv2 + ReLU of the IMPALA Atari network (examples/IMPALA/atari_model.py:59-71)
    for uint8 observations [n,4,42,42], as ONE fused MFMA kernel (inference only).  Returns f32
    [n, 3872] = the NCHW-flattened [n,32,11,11] activation.  `obs` may be a RingObservation (the actors' step).
    packed: atari42_conv12_pack(conv1_weight, conv2_weight) of the CURRENT weights (same result, faster start).
    save_a1 (needs `packed`, a materialised obs): returns (out, a1) with a1 f32 [n, 10000] = every observation's
    zero-padded conv1 tile, for atari42_conv12_backward(a1=...) — the learner's pair (parlhip_atari42_conv12_packed_save_u8_f32)."""
    if tuple(conv1_weight.shape) != (16, 4, 4, 4) or tuple(conv2_weight.shape) != (32, 16, 4, 4):
        raise N.ParlHipError('atari42_conv12: weights must be [16,4,4,4] and [32,16,4,4]')
    if packed is not None and (packed.dtype != torch.float32 or not packed.is_contiguous() or
                               packed.numel() * 4 != N.lib().parlhip_atari42_conv12_weights_bytes()):
        raise N.ParlHipError('atari42_conv12: packed must be atari42_conv12_pack\'s buffer')
    b1, b2 = _f32(conv1_bias.detach(), 'conv1_bias'), _f32(conv2_bias.detach(), 'conv2_bias')
    if isinstance(obs, RingObservation):
        if obs.dim != 42:
            raise N.ParlHipError('atari42_conv12: a 42x42 ring')
        if save_a1:
            raise N.ParlHipError('atari42_conv12: save_a1 is the learner\'s form (a materialised batch)')
        S, E = obs.ring.shape[0], obs.ring.shape[1]
        if out is None:
            out = torch.empty((E, 32 * 11 * 11), dtype=torch.float32, device=obs.device)
        if packed is not None:
            N.check(
                N.lib().parlhip_atari42_conv12_ring_packed_u8_f32(N.ptr(obs.ring), N.ptr(obs.since), S, E, obs.slot,
                                                                 N.ptr(packed), N.ptr(b1), N.ptr(b2), N.ptr(out),
                                                                 N.stream_ptr()),
                'parlhip_atari42_conv12_ring_packed_u8_f32')
            return out
        w1, w2 = _f32(conv1_weight.detach(), 'conv1_weight'), _f32(conv2_weight.detach(), 'conv2_weight')
        N.check(
            N.lib().parlhip_atari42_conv12_ring_u8_f32(N.ptr(obs.ring), N.ptr(obs.since), S, E, obs.slot, N.ptr(w1),
                                                      N.ptr(b1), N.ptr(w2), N.ptr(b2), N.ptr(out), N.stream_ptr()),
            'parlhip_atari42_conv12_ring_u8_f32')
        return out
    if obs.dtype != torch.uint8 or obs.dim() != 4 or tuple(obs.shape[1:]) != (4, 42, 42):
        raise N.ParlHipError('atari42_conv12: obs must be uint8 [n,4,42,42]')
    n = obs.shape[0]
    if out is None:
        out = torch.empty((n, 32 * 11 * 11), dtype=torch.float32, device=obs.device)
    if save_a1:
        if packed is None:
            raise N.ParlHipError('atari42_conv12: save_a1 needs the packed weights')
        a1 = torch.empty((n, N.lib().parlhip_atari42_conv12_a1_bytes(1) // 4), dtype=torch.float32, device=obs.device)
        N.check(
            N.lib().parlhip_atari42_conv12_packed_save_u8_f32(N.ptr(obs.contiguous()), N.ptr(packed), N.ptr(b1), N.ptr(b2),
                                                             N.ptr(out), N.ptr(a1), n, N.stream_ptr()),
            'parlhip_atari42_conv12_packed_save_u8_f32')
        return out, a1
    if packed is not None:
        N.check(
            N.lib().parlhip_atari42_conv12_packed_u8_f32(N.ptr(obs.contiguous()), N.ptr(packed), N.ptr(b1), N.ptr(b2),
                                                        N.ptr(out), n, N.stream_ptr()),
            'parlhip_atari42_conv12_packed_u8_f32')
        return out
    w1, w2 = _f32(conv1_weight.detach(), 'conv1_weight'), _f32(conv2_weight.detach(), 'conv2_weight')
    N.check(
        N.lib().parlhip_atari42_conv12_u8_f32(N.ptr(obs.contiguous()), N.ptr(w1), N.ptr(b1), N.ptr(w2), N.ptr(b2),
                                             N.ptr(out), n, N.stream_ptr()), 'parlhip_atari42_conv12_u8_f32')
    return out


def atari42_conv12_backward(obs, conv1_weight, conv1_bias, conv2_weight, a2, grad_a2, packed=None, a1=None):
    """Gradient of atari42_conv12 w.r.t. its four parameters given its output a2 [n,3872] and
    d loss / d a2 (the learner side of examples/IMPALA/atari_model.py:59-71; the observations get
    no gradient).  conv1 is recomputed inside the kernel — or, with `a1` (the forward's save_a1 output, needs
    `packed`), read back; deterministic.  Returns
    (d conv1_weight [16,4,4,4], d conv1_bias [16], d conv2_weight [32,16,4,4], d conv2_bias [32])."""
    if obs.dtype != torch.uint8 or obs.dim() != 4 or tuple(obs.shape[1:]) != (4, 42, 42):
        raise N.ParlHipError('atari42_conv12_backward: obs must be uint8 [n,4,42,42]')
    n = obs.shape[0]
    a2, grad_a2 = _f32(a2, 'a2'), _f32(grad_a2.contiguous(), 'grad_a2')
    if a2.numel() != n * 3872 or grad_a2.numel() != n * 3872:
        raise N.ParlHipError('atari42_conv12_backward: a2 / grad_a2 must be [n,3872]')
    dev = obs.device
    dw1 = torch.empty((16, 4, 4, 4), dtype=torch.float32, device=dev)
    db1 = torch.empty(16, dtype=torch.float32, device=dev)
    dw2 = torch.empty((32, 16, 4, 4), dtype=torch.float32, device=dev)
    db2 = torch.empty(32, dtype=torch.float32, device=dev)
    nb = N.lib().parlhip_atari42_conv12_bwd_workspace_bytes(n)
    ws = torch.empty(max(nb // 4, 1), dtype=torch.float32, device=dev)
    w1, b1 = _f32(conv1_weight.detach(), 'conv1_weight'), _f32(conv1_bias.detach(), 'conv1_bias')
    w2 = _f32(conv2_weight.detach(), 'conv2_weight')
    if a1 is not None:
        if packed is None or a1.dtype != torch.float32 or not a1.is_contiguous() or \
                a1.numel() * 4 != N.lib().parlhip_atari42_conv12_a1_bytes(n):
            raise N.ParlHipError('atari42_conv12_backward: a1 must be the forward\'s save_a1 output (and needs `packed`)')
        N.check(
            N.lib().parlhip_atari42_conv12_bwd_saved_f32(N.ptr(obs.contiguous()), N.ptr(packed), N.ptr(b1), N.ptr(a1),
                                                        N.ptr(a2), N.ptr(grad_a2), n, N.ptr(ws), N.ptr(dw1), N.ptr(db1),
                                                        N.ptr(dw2), N.ptr(db2), N.stream_ptr()),
            'parlhip_atari42_conv12_bwd_saved_f32')
        return dw1, db1, dw2, db2
    if packed is not None:   # atari42_conv12_pack of the weights the forward ran with (they have not changed since)
        N.check(
            N.lib().parlhip_atari42_conv12_bwd_packed_f32(N.ptr(obs.contiguous()), N.ptr(packed), N.ptr(b1), N.ptr(a2),
                                                         N.ptr(grad_a2), n, N.ptr(ws), N.ptr(dw1), N.ptr(db1),
                                                         N.ptr(dw2), N.ptr(db2), N.stream_ptr()),
            'parlhip_atari42_conv12_bwd_packed_f32')
        return dw1, db1, dw2, db2
    N.check(
        N.lib().parlhip_atari42_conv12_bwd_f32(N.ptr(obs.contiguous()), N.ptr(w1), N.ptr(b1), N.ptr(w2), N.ptr(a2),
                                              N.ptr(grad_a2), n, N.ptr(ws), N.ptr(dw1), N.ptr(db1), N.ptr(dw2),
                                              N.ptr(db2), N.stream_ptr()), 'parlhip_atari42_conv12_bwd_f32')
    return dw1, db1, dw2, db2


class Atari42Conv12Fn(torch.autograd.Function):
    """autograd node: forward = the fused conv1 + conv2 MFMA kernel on uint8 observations,
    backward = atari42_conv12_backward (one kernel + a fixed-order reduction)."""

    @staticmethod
    def forward(ctx, obs, w1, b1, w2, b2):
        # the learner's weights change with every update: the operand-order copy is made per call (one 9,216-thread
        # launch, also inside a captured update) — cheaper than fetching the operands scattered in every workgroup
        pk = atari42_conv12_pack(w1, w2) if obs.shape[0] >= 256 else None
        # batches of the reference's learner size keep the conv1 activation (40 KB per row) for the backward kernel,
        # which then loads it instead of recomputing conv1; a whole-rollout pass (51,200 rows = 2 GB) recomputes
        save = pk is not None and obs.shape[0] <= A1_SAVE_MAX_ROWS and os.environ.get('PARL_AMD_SAVE_A1', '1') != '0'
        if save:
            a2, a1 = atari42_conv12(obs, w1, b1, w2, b2, packed=pk, save_a1=True)
        else:
            a2, a1 = atari42_conv12(obs, w1, b1, w2, b2, packed=pk), None
        ctx.save_for_backward(obs, w1, b1, w2, a2)
        ctx.packed, ctx.a1 = pk, a1   # (not saved tensors: internal buffers nobody else writes)
        return a2

    @staticmethod
    def backward(ctx, grad_a2):
        obs, w1, b1, w2, a2 = ctx.saved_tensors
        dw1, db1, dw2, db2 = atari42_conv12_backward(obs, w1, b1, w2, a2, grad_a2, packed=ctx.packed, a1=ctx.a1)
        ctx.a1 = None
        return None, dw1, db1, dw2, db2


def atari84_conv1(obs, conv1_weight, conv1_bias, out=None, wt1=None):
    """conv1 + ReLU of the A2C Atari network (examples/A2C/atari_model.py:21-104: 4->32 k8 s4 p1,
    84x84 -> 20x20) for uint8 observations [n,4,84,84] as one MFMA kernel with the /255 fused
    (inference only).  Returns f32 [n,32,20,20].  `obs` may be a RingObservation (the actors' step).
    The kernel gets the weights in operand order (wt1 = _mfma_b_layout of the [32, 256] matrix; cached per weight
    tensor and version like atari84_conv23's): the scattered fetch from the nn.Conv2d layout was the start-up
    cost of every workgroup."""
    if tuple(conv1_weight.shape) != (32, 4, 8, 8) or tuple(conv1_bias.shape) != (32, ):
        raise N.ParlHipError('atari84_conv1: weight must be [32,4,8,8], bias [32]')
    b1 = _f32(conv1_bias.detach(), 'conv1_bias')
    if not conv1_weight.is_cuda:
        N.ptr(conv1_weight)   # raises: no CPU path
    # (rebuilt when the tensor's version moved — every optimizer step for the learner, once per rollout for the actors —
    # and always for parameters a graph replay writes; wt1: the caller's own operand-order copy of the CURRENT weights)
    if wt1 is None and (torch.is_grad_enabled() or torch.cuda.is_current_stream_capturing()):
        # the learner (weights change every update) and captures (a layout built OUTSIDE the graph would be baked in
        # and go stale on replay; the cache would also record an event inside the capture): built inline, as
        # atari84_conv23 does
        wt1 = atari84_conv1_layout(conv1_weight)
    elif wt1 is None:
        wt1 = _cached_layout(conv1_weight, 'wt1', lambda w: atari84_conv1_layout(w))
    elif wt1.dtype != torch.float32 or wt1.numel() != 64 * 2 * 64 or not wt1.is_contiguous():
        raise N.ParlHipError('atari84_conv1: wt1 must be atari84_conv1_layout\'s buffer')
    if isinstance(obs, RingObservation):
        if obs.dim != 84:
            raise N.ParlHipError('atari84_conv1: an 84x84 ring')
        S, E = obs.ring.shape[0], obs.ring.shape[1]
        if out is None:
            out = torch.empty((E, 32, 20, 20), dtype=torch.float32, device=obs.device)
        elif out.dtype != torch.float32 or out.numel() != E * 12800 or not out.is_contiguous():
            raise N.ParlHipError('atari84_conv1: out must be contiguous f32 [n,32,20,20]')
        N.check(
            N.lib().parlhip_atari84_conv1_ring_packed_u8_f32(N.ptr(obs.ring), N.ptr(obs.since), S, E, obs.slot,
                                                            N.ptr(wt1), N.ptr(b1), N.ptr(out), N.stream_ptr()),
            'parlhip_atari84_conv1_ring_packed_u8_f32')
        return out
    if obs.dtype != torch.uint8 or obs.dim() != 4 or tuple(obs.shape[1:]) != (4, 84, 84):
        raise N.ParlHipError('atari84_conv1: obs must be uint8 [n,4,84,84]')
    n = obs.shape[0]
    if out is None:
        out = torch.empty((n, 32, 20, 20), dtype=torch.float32, device=obs.device)
    elif out.dtype != torch.float32 or out.numel() != n * 12800 or not out.is_contiguous():
        raise N.ParlHipError('atari84_conv1: out must be contiguous f32 [n,32,20,20]')
    N.check(
        N.lib().parlhip_atari84_conv1_packed_u8_f32(N.ptr(obs.contiguous()), N.ptr(wt1), N.ptr(b1), N.ptr(out), n,
                                                   N.stream_ptr()), 'parlhip_atari84_conv1_packed_u8_f32')
    return out


def _mfma_b_layout(wflat):
    """[N, K] weight matrix (N = 16*NT, K = 4*KS) -> [KS, NT, 64] in MFMA B-operand order:
    out[ks][nt][q*16 + col] = w[16*nt + col][4*ks + q] (one coalesced 256-byte load per wave)"""
    n, k = wflat.shape
    return wflat.reshape(n // 16, 16, k // 4, 4).permute(2, 0, 3, 1).contiguous()


_layout_cache = {}


def _cached_layout(w, kind, fn):
    """MFMA operand-order copy of a weight tensor, rebuilt only when the tensor was written since (its autograd
    version counter moves on every eager in-place write: optimizer steps, the actors' snapshot copy).  The actors
    call the forward kernels once per env step with weights that change once per rollout: without the cache
    every step re-laid out both matrices (three small copy kernels and their host time).
    Two things the version counter cannot see are handled explicitly: (a) a hipGraph replay writes parameters
    without touching the counter — GraphedLearn flags the parameters it updates (`_parl_graph_written`) and
    flagged tensors are never cached; (b) the copy is built on ONE stream — an event recorded behind the
    build is kept with it and a consumer on another stream waits for it (several env groups on their own
    actor streams share one actor model)."""
    if getattr(w, '_parl_graph_written', False):
        return fn(w)
    key = (id(w), kind)
    hit = _layout_cache.get(key)
    ver = w._version
    cur = torch.cuda.current_stream(w.device)
    if hit is not None and hit[0] is w and hit[1] == ver and hit[2] == w.data_ptr():
        if hit[5] != cur:
            cur.wait_event(hit[4])
            hit[3].record_stream(cur)   # the allocator must not hand the block out again while this stream reads it
        return hit[3]
    out = fn(w)
    ev = torch.cuda.Event()
    ev.record(cur)
    if len(_layout_cache) > 64:
        _layout_cache.clear()
    _layout_cache[key] = (w, ver, w.data_ptr(), out, ev, cur)
    return out


def atari84_conv1_layout(conv1_weight):
    """conv1's weight matrix in MFMA operand order: f32 [64, 2, 4, 16] (= [ks][nt][lane])"""
    return _mfma_b_layout(_f32(conv1_weight.detach(), 'conv1_weight').reshape(32, 256))


def atari84_conv23_layouts(conv2_weight, conv3_weight):
    """(wt2 f32 [128, 4, 4, 16], wt3 f32 [144, 4, 4, 16]): conv2 / conv3 weights in MFMA operand order ([ks][nt][lane])"""
    return (_mfma_b_layout(_f32(conv2_weight.detach(), 'conv2_weight').reshape(64, 512)),
            _mfma_b_layout(_f32(conv3_weight.detach(), 'conv3_weight').permute(0, 2, 3, 1).reshape(64, 576)))   # k' = (kh*3 + kw)*64 + c


def atari84_conv23(a1, conv2_weight, conv2_bias, conv3_weight, conv3_bias, save_a2=False, wt23=None):
    """conv2 + ReLU + conv3 + ReLU of the A2C Atari network (examples/A2C/atari_model.py:21-104) fused
    in one MFMA kernel: a1 f32 [n,32,20,20] (atari84_conv1's output) -> a3 f32 [n,5184]; with
    save_a2 also the conv2 activation [n,64,11,11] (for the backward pass)."""
    if a1.dtype != torch.float32 or a1.dim() != 4 or tuple(a1.shape[1:]) != (32, 20, 20):
        raise N.ParlHipError('atari84_conv23: a1 must be f32 [n,32,20,20]')
    if tuple(conv2_weight.shape) != (64, 32, 4, 4) or tuple(conv3_weight.shape) != (64, 64, 3, 3):
        raise N.ParlHipError('atari84_conv23: weights must be [64,32,4,4] and [64,64,3,3]')
    n = a1.shape[0]
    w2 = _f32(conv2_weight.detach(), 'conv2_weight')
    w3 = _f32(conv3_weight.detach(), 'conv3_weight')
    if wt23 is not None:   # the caller's own operand-order copies of the CURRENT weights (atari84_conv23_layouts)
        wt2, wt3 = wt23
        if wt2.numel() != 64 * 512 or wt3.numel() != 64 * 576 or not (wt2.is_contiguous() and wt3.is_contiguous()):
            raise N.ParlHipError('atari84_conv23: wt23 must be atari84_conv23_layouts\' pair')
    elif torch.is_grad_enabled() or torch.cuda.is_current_stream_capturing():
        wt2 = _mfma_b_layout(w2.reshape(64, 512))
        wt3 = _mfma_b_layout(w3.permute(0, 2, 3, 1).reshape(64, 576))   # k' = (kh*3 + kw)*64 + c
    else:  # the actors: same weights for a whole rollout
        wt2 = _cached_layout(conv2_weight, 'wt2', lambda w: _mfma_b_layout(w.detach().reshape(64, 512)))
        wt3 = _cached_layout(conv3_weight, 'wt3',
                             lambda w: _mfma_b_layout(w.detach().permute(0, 2, 3, 1).reshape(64, 576)))
    b2, b3 = _f32(conv2_bias.detach(), 'conv2_bias'), _f32(conv3_bias.detach(), 'conv3_bias')
    a3 = torch.empty((n, 64 * 81), dtype=torch.float32, device=a1.device)
    a2 = torch.empty((n, 64, 11, 11), dtype=torch.float32, device=a1.device) if save_a2 else None
    N.check(
        N.lib().parlhip_atari84_conv23_f32(N.ptr(a1.contiguous()), N.ptr(wt2), N.ptr(b2), N.ptr(wt3), N.ptr(b3),
                                          N.ptr(a2) if a2 is not None else None, N.ptr(a3), n, N.stream_ptr()),
        'parlhip_atari84_conv23_f32')
    return (a3, a2) if save_a2 else a3


def atari84_conv3_backward(a2, a3, grad_a3, conv3_weight):
    """backward of conv3 + ReLU of the A2C Atari network: (dz2 [n,64,11,11] = grad w.r.t. a2 masked by
    a2 > 0, d conv3.weight [64,64,3,3], d conv3.bias [64]); deterministic"""
    n = a2.shape[0]
    a2, a3 = _f32(a2, 'a2'), _f32(a3, 'a3')
    g3 = _f32(grad_a3.contiguous(), 'grad_a3')
    if a2.numel() != n * 7744 or a3.numel() != n * 5184 or g3.numel() != n * 5184:
        raise N.ParlHipError('atari84_conv3_backward: a2 [n,64,11,11], a3 / grad_a3 [n,5184]')
    w3 = _f32(conv3_weight.detach(), 'conv3_weight')
    wt3b = _mfma_b_layout(w3.permute(1, 2, 3, 0).reshape(64, 576))   # [c][k'' = tap*64 + o]
    dev = a2.device
    dz2 = torch.empty((n, 64, 11, 11), dtype=torch.float32, device=dev)
    out = torch.empty(64 * 576 + 64, dtype=torch.float32, device=dev)
    ws = torch.empty(max(N.lib().parlhip_atari84_conv3_bwd_workspace_bytes(n) // 4, 1), dtype=torch.float32, device=dev)
    N.check(
        N.lib().parlhip_atari84_conv3_bwd_f32(N.ptr(a2.contiguous()), N.ptr(a3.contiguous()), N.ptr(g3), N.ptr(wt3b), n,
                                             N.ptr(ws), N.ptr(dz2), N.ptr(out), N.stream_ptr()),
        'parlhip_atari84_conv3_bwd_f32')
    dw3 = out[:64 * 576].view(64, 9, 64).permute(0, 2, 1).reshape(64, 64, 3, 3).contiguous()
    return dz2, dw3, out[64 * 576:].clone()


def atari84_conv2_backward(a1, dz2, conv2_weight):
    """backward of conv2 + ReLU: (dz1 [n,32,20,20] = grad w.r.t. a1 masked by a1 > 0,
    d conv2.weight [64,32,4,4], d conv2.bias [64]); deterministic"""
    n = a1.shape[0]
    a1, dz2 = _f32(a1, 'a1'), _f32(dz2, 'dz2')
    if a1.numel() != n * 12800 or dz2.numel() != n * 7744:
        raise N.ParlHipError('atari84_conv2_backward: a1 [n,32,20,20], dz2 [n,64,11,11]')
    w2 = _f32(conv2_weight.detach(), 'conv2_weight')
    # per output parity class (py, px): B[k = (o, a, b)][c] = w2[o][c][py + 2a][px + 2b]
    wt2b = torch.stack([_mfma_b_layout(w2[:, :, py::2, px::2].permute(1, 0, 2, 3).reshape(32, 256))
                        for py in (0, 1) for px in (0, 1)]).contiguous()
    dev = a1.device
    dz1 = torch.empty((n, 32, 20, 20), dtype=torch.float32, device=dev)
    out = torch.empty(64 * 512 + 64, dtype=torch.float32, device=dev)
    ws = torch.empty(max(N.lib().parlhip_atari84_conv2_bwd_workspace_bytes(n) // 4, 1), dtype=torch.float32, device=dev)
    N.check(
        N.lib().parlhip_atari84_conv2_bwd_f32(N.ptr(a1.contiguous()), N.ptr(dz2.contiguous()), N.ptr(wt2b), n, N.ptr(ws),
                                             N.ptr(dz1), N.ptr(out), N.stream_ptr()), 'parlhip_atari84_conv2_bwd_f32')
    return dz1, out[:64 * 512].view(64, 32, 4, 4), out[64 * 512:]


def atari84_conv1_backward(obs, dz1):
    """backward of conv1 w.r.t. its parameters: (d conv1.weight [32,4,8,8], d conv1.bias [32]); deterministic"""
    if obs.dtype != torch.uint8 or obs.dim() != 4 or tuple(obs.shape[1:]) != (4, 84, 84):
        raise N.ParlHipError('atari84_conv1_backward: obs must be uint8 [n,4,84,84]')
    n = obs.shape[0]
    dz1 = _f32(dz1, 'dz1')
    if dz1.numel() != n * 12800:
        raise N.ParlHipError('atari84_conv1_backward: dz1 must be [n,32,20,20]')
    dev = obs.device
    out = torch.empty(32 * 256 + 32, dtype=torch.float32, device=dev)
    ws = torch.empty(max(N.lib().parlhip_atari84_conv1_bwd_workspace_bytes(n) // 4, 1), dtype=torch.float32, device=dev)
    N.check(
        N.lib().parlhip_atari84_conv1_bwd_f32(N.ptr(obs.contiguous()), N.ptr(dz1.contiguous()), n, N.ptr(ws), N.ptr(out),
                                             N.stream_ptr()), 'parlhip_atari84_conv1_bwd_f32')
    return out[:32 * 256].view(32, 4, 8, 8), out[32 * 256:]


class Atari84TrunkFn(torch.autograd.Function):
    """autograd node for the three convolutions of the A2C Atari network on uint8 observations:
    forward = conv1 (MFMA, u8 -> a1) + fused conv2/conv3 (a2, a3), backward = three per-layer MFMA
    kernels (conv3: dW3 + dz2; conv2: dW2 + dz1; conv1: dW1).  No im2col, deterministic; the
    observations get no gradient."""

    @staticmethod
    def forward(ctx, obs, w1, b1, w2, b2, w3, b3):
        a1 = atari84_conv1(obs, w1, b1)
        a3, a2 = atari84_conv23(a1, w2, b2, w3, b3, save_a2=True)
        ctx.save_for_backward(obs, a1, a2, a3, w2, w3)
        return a3

    @staticmethod
    def backward(ctx, grad_a3):
        obs, a1, a2, a3, w2, w3 = ctx.saved_tensors
        dz2, dw3, db3 = atari84_conv3_backward(a2, a3, grad_a3, w3)
        dz1, dw2, db2 = atari84_conv2_backward(a1, dz2, w2)
        dw1, db1 = atari84_conv1_backward(obs, dz1)
        return None, dw1, db1, dw2, db2, dw3, db3


def _f64(t, name):
    if t.dtype != torch.float64 or not t.is_cuda:
        raise N.ParlHipError('%s must be a float64 CUDA tensor' % name)
    if not t.is_contiguous():
        raise N.ParlHipError('%s must be contiguous (it is updated in place)' % name)
    return t


def vecnorm_obs(raw, mean, var, count, mask=None, out=None, clipob=10.0, eps=1e-8, update=True, out64=None):
    """VecNormalizeEnv._obfilt (parl/env/mujoco_wrappers.py:140-156) for E envs, each with its own
    RunningMeanStd: raw f64 [E,D]; mean/var f64 [E,D] and count f64 [E] updated IN PLACE when
    `update`; mask (bool/uint8 [E]) restricts the call to some envs (reset path).  Returns the
    float32 normalised observations [E,D] (written into `out` when given)."""
    raw = _f64(raw.contiguous(), 'raw')
    E, D = raw.shape
    _f64(mean, 'mean'), _f64(var, 'var'), _f64(count, 'count')
    if tuple(mean.shape) != (E, D) or tuple(var.shape) != (E, D) or count.numel() != E:
        raise N.ParlHipError('vecnorm_obs: mean/var must be [E,D], count [E]')
    if mask is not None:
        mask = (mask.contiguous().view(torch.uint8) if mask.dtype == torch.bool else mask.contiguous())
        if mask.dtype != torch.uint8 or mask.numel() != E:
            raise N.ParlHipError('vecnorm_obs: mask must be bool/uint8 [E]')
    if out is None:
        out = torch.zeros((E, D), dtype=torch.float32, device=raw.device)
    elif out.dtype != torch.float32 or tuple(out.shape) != (E, D) or not out.is_contiguous():
        raise N.ParlHipError('vecnorm_obs: out must be contiguous f32 [E,D]')
    if out64 is not None:
        _f64(out64, 'out64')
    N.check(
        N.lib().parlhip_vecnorm_obs_f64(N.ptr(raw), N.ptr(mean), N.ptr(var), N.ptr(count), N.ptr(mask), N.ptr(out),
                                       N.ptr(out64), E, D, float(clipob), float(eps), 1 if update else 0,
                                       N.stream_ptr()), 'parlhip_vecnorm_obs_f64')
    return out


def vecnorm_reward(rew, done, ret, ret_mean, ret_var, ret_count, gamma=0.99, cliprew=10.0, eps=1e-8, out=None,
                   out64=None):
    """VecNormalizeEnv.step's reward half (parl/env/mujoco_wrappers.py:120-136); the four running
    arrays (f64 [E]) are updated IN PLACE.  Returns the float32 normalised rewards [E]."""
    rew = _f64(rew.contiguous(), 'rew')
    E = rew.numel()
    done = done.contiguous().view(torch.uint8) if done.dtype == torch.bool else done.contiguous()
    if done.dtype != torch.uint8 or done.numel() != E:
        raise N.ParlHipError('vecnorm_reward: done must be bool/uint8 [E]')
    for t, n in ((ret, 'ret'), (ret_mean, 'ret_mean'), (ret_var, 'ret_var'), (ret_count, 'ret_count')):
        if _f64(t, n).numel() != E:
            raise N.ParlHipError('vecnorm_reward: %s must be [E]' % n)
    if out is None:
        out = torch.empty(E, dtype=torch.float32, device=rew.device)
    if out64 is not None:
        _f64(out64, 'out64')
    N.check(
        N.lib().parlhip_vecnorm_reward_f64(N.ptr(rew), N.ptr(done), N.ptr(ret), N.ptr(ret_mean), N.ptr(ret_var),
                                          N.ptr(ret_count), N.ptr(out), N.ptr(out64), E, float(gamma),
                                          float(cliprew), float(eps), N.stream_ptr()), 'parlhip_vecnorm_reward_f64')
    return out


def ppo_sample_batch(obs, actions, logprobs, advantages, returns, values, idx):
    """RolloutStorage.sample_batch (examples/PPO/storage.py:66-76): gather the six flattened rollout
    arrays by one minibatch index in ONE launch.  obs [N,...] / actions [N,...] f32, the others
    f32 [N], idx int64 [M].  Returns (obs, actions, logprobs, advantages, returns, values)[idx]."""
    lp, adv, ret, val = [_f32(x, n).reshape(-1) for x, n in ((logprobs, 'logprobs'), (advantages, 'advantages'),
                                                             (returns, 'returns'), (values, 'values'))]
    n = lp.numel()
    obs, actions = _f32(obs, 'obs'), _f32(actions, 'actions')
    if obs.shape[0] != n or actions.shape[0] != n or adv.numel() != n or ret.numel() != n or val.numel() != n:
        raise N.ParlHipError('ppo_sample_batch: all arrays must have the same leading size')
    if idx.dtype != torch.int64:
        raise N.ParlHipError('idx must be int64')
    idx = idx.contiguous().reshape(-1)
    m = idx.numel()
    do = obs.numel() // n if n else 0
    da = actions.numel() // n if n else 0
    dev = lp.device
    o_obs = torch.empty((m, ) + tuple(obs.shape[1:]), dtype=torch.float32, device=dev)
    o_act = torch.empty((m, ) + tuple(actions.shape[1:]), dtype=torch.float32, device=dev)
    o4 = [torch.empty(m, dtype=torch.float32, device=dev) for _ in range(4)]
    N.check(
        N.lib().parlhip_ppo_sample_batch_f32(N.ptr(obs), N.ptr(actions), N.ptr(lp), N.ptr(adv), N.ptr(ret), N.ptr(val),
                                            N.ptr(idx), N.ptr(o_obs), N.ptr(o_act), N.ptr(o4[0]), N.ptr(o4[1]),
                                            N.ptr(o4[2]), N.ptr(o4[3]), n, m, do, da, N.stream_ptr()),
        'parlhip_ppo_sample_batch_f32')
    return (o_obs, o_act) + tuple(o4)


def consume_device_errors():
    """Synchronise and return/clear the device-side data-error flag (bad action index)."""
    return N.check(N.lib().parlhip_consume_device_errors(N.stream_ptr()),
                   'parlhip_consume_device_errors')
