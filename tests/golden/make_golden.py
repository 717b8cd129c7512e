"""Generate the golden fixtures under tests/golden/ FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference, read-only).  Nothing is copied from
the reference into this repository: the reference's own functions are imported / exec'd in
place and only their numeric inputs and outputs are stored (small .npz files).

    python tests/golden/make_golden.py

Sources (paths relative to /root/reference):
  * vtrace_known_answer.npz — the known-answer recipe and the O(T^2) numpy ground truth of
    parl/algorithms/paddle/impala/tests/vtrace_test_paddle.py:28-113 (its two pure-numpy
    helpers are exec'd from the file; the paddle half of that test cannot run here).
  * calc_gae.npz — parl/utils/rl_utils.py:21-51 (calc_gae, calc_discount_sum_rewards) run
    on the per-segment inputs examples/A2C/actor.py:73-85 would build.
  * ppo_compute_returns.npz — examples/PPO/storage.py:18-64 (RolloutStorage.compute_returns).
  * scheduler.npz — parl/utils/scheduler.py (PiecewiseScheduler, LinearDecayScheduler).
"""
import ast
import collections
import importlib.util
import os
import sys

sys.dont_write_bytecode = True  # modules are imported from /root/reference by path: never leave a __pycache__ there

import numpy as np

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))


def _load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _exec_functions(path, names, extra_globals):
    """exec only the named top-level functions of a reference file (skips its imports)."""
    src = open(path).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(keep) == len(names), (path, names)
    g = dict(extra_globals)
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, 'exec'), g)
    return g


def make_vtrace():
    path = os.path.join(REF, 'parl/algorithms/paddle/impala/tests/vtrace_test_paddle.py')
    VTraceReturns = collections.namedtuple('VTraceReturns', ['vs', 'pg_advantages'])
    fake_vtrace = collections.namedtuple('m', ['VTraceReturns'])(VTraceReturns)
    g = _exec_functions(path, ['_shaped_arange', '_ground_truth_calculation'],
                        {'np': np, 'vtrace': fake_vtrace})
    sa, gt = g['_shaped_arange'], g['_ground_truth_calculation']
    out = {}
    # the reference test's own two cases (vtrace_test_paddle.py:78-113) ...
    cases = [('ref_B1', 5, 1, 3.7, 2.2), ('ref_B4', 5, 4, 3.7, 2.2)]
    # ... plus the same recipe at other sizes / thresholds (incl. the IMPALA config's 1.0/1.0
    # and a no-clip case, which the ground truth treats as `falsy -> unclipped`)
    cases += [('B7_T13', 13, 7, 1.0, 1.0), ('B3_T50', 50, 3, 1.0, 1.0),
              ('B2_T9_noclip', 9, 2, None, None)]
    for name, seq_len, batch_size, crho, cpg in cases:
        log_rhos = sa(seq_len, batch_size) / (batch_size * seq_len)
        log_rhos = 5 * (log_rhos - 0.5)
        values = {
            'behaviour_actions_log_probs': np.ones(log_rhos.shape, dtype='float32'),
            'target_actions_log_probs': log_rhos + 1.0,
            'discounts': np.array([[0.9 / (b + 1) for b in range(batch_size)]
                                   for _ in range(seq_len)], dtype=np.float32),
            'rewards': sa(seq_len, batch_size),
            'values': sa(seq_len, batch_size) / batch_size,
            'bootstrap_value': sa(batch_size) + 1.0,
            'clip_rho_threshold': crho,
            'clip_pg_rho_threshold': cpg,
        }
        res = gt(**values)
        for k, v in values.items():
            if k.startswith('clip'):
                out['%s/%s' % (name, k)] = np.array(np.nan if v is None else v, np.float32)
            else:
                out['%s/%s' % (name, k)] = np.asarray(v, np.float32)
        out['%s/vs' % name] = np.asarray(res.vs, np.float32)
        out['%s/pg_advantages' % name] = np.asarray(res.pg_advantages, np.float32)
    np.savez(os.path.join(OUT, 'vtrace_known_answer.npz'), **out)
    print('vtrace ref_B1 vs[:,0] =', out['ref_B1/vs'][:, 0])
    print('vtrace ref_B4 vs[:,3] =', out['ref_B4/vs'][:, 3])


def make_calc_gae():
    rl = _load_by_path('ref_rl_utils', os.path.join(REF, 'parl/utils/rl_utils.py'))
    out = {}
    # the two samples recorded in SURVEY.md §8c
    s1 = rl.calc_gae([1, 0, -1, 1], np.array([.5, .25, -.5, .75], np.float32),
                     np.array([0.3], np.float32), 0.99, 1.0)
    s2 = rl.calc_gae([1, 0, -1, 1], np.array([.5, .25, -.5, .75], np.float32), 0, 0.99, 0.95)
    out['sample1'] = np.asarray(s1, np.float64)
    out['sample2'] = np.asarray(s2, np.float64)
    # A2C-actor style rollouts: [T,B] with dones; per (env, segment) reference calls exactly
    # as examples/A2C/actor.py:73-93 makes them (values are np.float32 scalars in lists,
    # rewards np.sign() float64 scalars, next_value 0 or a float32 [1] array).
    rng = np.random.default_rng(1234)
    for name, T, B, lam, pdone in [('a2c_T20_B6_lam1', 20, 6, 1.0, 0.1),
                                   ('a2c_T20_B6_lam95', 20, 6, 0.95, 0.1),
                                   ('a2c_T5_B3_lam1', 5, 3, 1.0, 0.4),
                                   ('a2c_T128_B4_lam9', 128, 4, 0.9, 0.02)]:
        rewards = np.sign(rng.integers(-1, 2, size=(T, B)).astype(np.float64))
        values = rng.standard_normal((T, B)).astype(np.float32)
        dones = rng.random((T, B)) < pdone
        next_value = rng.standard_normal(B).astype(np.float32)
        adv = np.zeros((T, B), np.float64)
        tgt = np.zeros((T, B), np.float64)
        for b in range(B):
            seg_r, seg_v, start = [], [], 0
            for t in range(T):
                seg_r.append(rewards[t, b])
                seg_v.append(values[t, b])
                if dones[t, b] or t == T - 1:
                    nv = 0
                    if not dones[t, b]:
                        nv = np.array([next_value[b]], np.float32)
                    a = rl.calc_gae(seg_r, seg_v, nv, 0.99, lam)
                    adv[start:t + 1, b] = a
                    tgt[start:t + 1, b] = a + seg_v
                    seg_r, seg_v, start = [], [], t + 1
        out[name + '/rewards'] = rewards.astype(np.float32)
        out[name + '/values'] = values
        out[name + '/dones'] = dones
        out[name + '/next_value'] = next_value
        out[name + '/lam'] = np.array(lam)
        out[name + '/advantages'] = adv
        out[name + '/target_values'] = tgt
    x = rng.standard_normal((17, 3))
    out['dsum/x'] = x.astype(np.float32)
    out['dsum/out'] = np.stack([rl.calc_discount_sum_rewards(x.astype(np.float32)[:, b], 0.97)
                                for b in range(3)], 1)
    np.savez(os.path.join(OUT, 'calc_gae.npz'), **out)
    print('calc_gae sample1 =', out['sample1'], ' sample2 =', out['sample2'])


def make_ppo():
    st = _load_by_path('ref_ppo_storage', os.path.join(REF, 'examples/PPO/storage.py'))
    Space = collections.namedtuple('Space', ['shape'])
    rng = np.random.default_rng(99)
    out = {}
    for name, T, E, g, lam in [('T16_E8', 16, 8, 0.99, 0.95), ('T64_E5', 64, 5, 0.99, 0.95),
                               ('T7_E3_g9_l1', 7, 3, 0.9, 1.0)]:
        rs = st.RolloutStorage(T, E, Space((3, )), Space((2, )))
        for t in range(T):
            rs.append(rng.standard_normal((E, 3)).astype(np.float32),
                      rng.standard_normal((E, 2)).astype(np.float32),
                      rng.standard_normal(E).astype(np.float32),
                      np.clip(rng.standard_normal(E), -10, 10).astype(np.float32),
                      (rng.random(E) < 0.15).astype(np.float32),
                      rng.standard_normal(E).astype(np.float32))
        value = rng.standard_normal(E).astype(np.float32)
        done = (rng.random(E) < 0.3).astype(np.float32)
        adv, ret = rs.compute_returns(value, done, gamma=g, gae_lambda=lam)
        out[name + '/rewards'] = rs.rewards
        out[name + '/values'] = rs.values
        out[name + '/dones'] = rs.dones
        out[name + '/value'] = value
        out[name + '/done'] = done
        out[name + '/gamma_lam'] = np.array([g, lam])
        out[name + '/advantages'] = adv
        out[name + '/returns'] = ret
        assert adv.dtype == np.float32
    np.savez(os.path.join(OUT, 'ppo_compute_returns.npz'), **out)
    print('ppo adv[0,:3] =', out['T16_E8/advantages'][0, :3])


def make_scheduler():
    # scheduler.py only needs `six`; load by path to avoid importing the parl package
    sch = _load_by_path('ref_scheduler', os.path.join(REF, 'parl/utils/scheduler.py'))
    out = {}
    p = sch.PiecewiseScheduler([(0, 0.001), (20000, 0.0005), (40000, 0.0001)])
    out['piecewise_step1'] = np.array([p.step() for _ in range(5)])
    p = sch.PiecewiseScheduler([(0, 0.001), (20, 0.0005), (40, 0.0001)])
    out['piecewise_step7'] = np.array([p.step(7) for _ in range(10)])
    l = sch.LinearDecayScheduler(0.001, 100)
    out['linear_step9'] = np.array([l.step(9) for _ in range(14)])
    np.savez(os.path.join(OUT, 'scheduler.npz'), **out)
    print('scheduler piecewise_step7 =', out['piecewise_step7'])


if __name__ == '__main__':
    if not os.path.isdir(REF):
        sys.exit('needs /root/reference (build container only)')
    make_vtrace()
    make_calc_gae()
    make_ppo()
    make_scheduler()
