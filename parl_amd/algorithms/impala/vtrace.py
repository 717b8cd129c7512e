"""vtrace.from_importance_weights with the reference's signature and return type
(parl/algorithms/paddle/impala/vtrace.py:31-139), computed by one HIP kernel launch instead of
the reference's per-time-step Python loop (vtrace.py:116-122)."""
import collections

import torch

from ... import ops

VTraceReturns = collections.namedtuple('VTraceReturns', ['vs', 'pg_advantages'])


@torch.no_grad()
def from_importance_weights(behaviour_actions_log_probs,
                            target_actions_log_probs,
                            discounts,
                            rewards,
                            values,
                            bootstrap_value,
                            clip_rho_threshold=1.0,
                            clip_pg_rho_threshold=1.0,
                            name='vtrace_from_logits'):
    """All inputs float32 [T,B] (bootstrap_value [B]) on the GPU; returns VTraceReturns(vs,
    pg_advantages), both [T,B], no gradient (vtrace.py:36 @paddle.no_grad)."""
    rank = len(behaviour_actions_log_probs.shape)  # vtrace.py:90-95
    assert len(target_actions_log_probs.shape) == rank
    assert len(values.shape) == rank
    assert len(bootstrap_value.shape) == (rank - 1)
    assert len(discounts.shape) == rank
    assert len(rewards.shape) == rank
    if rank != 2:
        T = behaviour_actions_log_probs.shape[0]
        shp = behaviour_actions_log_probs.shape
        f = lambda t: t.reshape(T, -1)
        vs, pg = ops.vtrace(f(behaviour_actions_log_probs), f(target_actions_log_probs), f(discounts), f(rewards),
                            f(values), bootstrap_value.reshape(-1), clip_rho_threshold, clip_pg_rho_threshold)
        return VTraceReturns(vs=vs.reshape(shp), pg_advantages=pg.reshape(shp))
    vs, pg = ops.vtrace(behaviour_actions_log_probs, target_actions_log_probs, discounts, rewards, values,
                        bootstrap_value, clip_rho_threshold, clip_pg_rho_threshold)
    return VTraceReturns(vs=vs, pg_advantages=pg)
