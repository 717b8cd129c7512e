"""torch-CPU baselines of BASELINE.md section 3 — TEST / MEASUREMENT INFRASTRUCTURE ONLY (imported by
bench.py's cpu_baseline leg and by tests; never by parl_amd).

  learner update   the reference's OWN torch A2C.learn (parl/algorithms/torch/a2c.py:40-81) on its own
                   ActorCritic (benchmark/torch/a2c/atari_model.py:23-104), staged byte for byte into
                   oracle/_ref/torch_alg/ by oracle/make_ref.py (build()) and loaded by path: kind "reference".
                   The two files import `parl` for three names (parl.Algorithm, parl.Model,
                   parl.utils.utils.check_model_method); a stub module supplies exactly those for the duration of
                   the import — the arithmetic that is timed is the reference's.
                   Batch: 500 rows of uint8 84x84x4 — the reference's A2C update (5 actors x 5 envs x 20 steps,
                   examples/A2C/a2c_config.py:22-30).
  V-trace          vtrace.py:99-137 as torch-CPU ops with the reference's per-t loop (kind "port": the
                   reference's is Paddle), at T'=49 x B=1024 (the reference shape) and T'=127 x B=65,536.
"""
import importlib.util
import os
import sys
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
STAGED = os.path.join(HERE, '_ref', 'torch_alg')


def _load_reference_a2c():
    import torch
    keep = {k: sys.modules.get(k) for k in ('parl', 'parl.utils', 'parl.utils.utils')}
    stub = types.ModuleType('parl')
    stub.Model = torch.nn.Module          # parl/core/torch/model.py:26 `class Model(nn.Module, ModelBase)`
    stub.Algorithm = object               # parl/core/torch/algorithm.py: attribute holder
    stub.utils = types.ModuleType('parl.utils')
    stub.utils.utils = types.ModuleType('parl.utils.utils')
    stub.utils.utils.check_model_method = lambda model, method, algo: getattr(model, method)
    sys.modules.update({'parl': stub, 'parl.utils': stub.utils, 'parl.utils.utils': stub.utils.utils})
    dwb, sys.dont_write_bytecode = sys.dont_write_bytecode, True
    try:
        mods = []
        for name in ('a2c', 'atari_model'):
            spec = importlib.util.spec_from_file_location('ref_torch_alg_' + name, os.path.join(STAGED, name + '.py'))
            m = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(m)
            mods.append(m)
        return mods[0].A2C, mods[1].ActorCritic
    finally:
        sys.dont_write_bytecode = dwb
        for k, v in keep.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def time_reference_a2c_learn(threads=8, rows=500, runs=5, act_dim=6):
    import torch
    if not os.path.exists(os.path.join(STAGED, 'a2c.py')):
        return {'error': 'oracle/_ref/torch_alg/ not staged (build() stages it where /root/reference exists)'}
    A2C, ActorCritic = _load_reference_a2c()
    n0 = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        torch.manual_seed(0)
        model = ActorCritic(act_dim)
        alg = A2C(model, {'vf_loss_coeff': 0.5, 'learning_rate': 0.001})
        rng = np.random.default_rng(0)
        obs = torch.from_numpy(rng.integers(0, 256, (rows, 4, 84, 84), dtype=np.uint8)).float()
        act = torch.from_numpy(rng.integers(0, act_dim, rows).astype(np.int64))
        adv = torch.from_numpy(rng.standard_normal(rows).astype(np.float32))
        tgt = torch.from_numpy(rng.standard_normal(rows).astype(np.float32))
        alg.learn(obs, act, adv, tgt, 1e-3, -0.01)  # warm-up
        ts = []
        for _ in range(runs):
            t0 = time.perf_counter()
            out = alg.learn(obs, act, adv, tgt, 1e-3, -0.01)
            ts.append(time.perf_counter() - t0)
        assert all(np.isfinite(float(x.detach())) for x in out)
        s = float(np.median(ts))
        return {'seconds_per_update': s, 'updates_per_s': 1.0 / s, 'rows_per_s': rows / s, 'rows': rows, 'kind': 'reference',
                'cores': threads,
                'what': 'parl/algorithms/torch/a2c.py A2C.learn on benchmark/torch/a2c/atari_model.py ActorCritic, '
                        '500 rows of 4x84x84 (the reference\'s A2C update), torch-CPU'}
    finally:
        torch.set_num_threads(n0)


def vtrace_torch(blp, tlp, discounts, rewards, values, bootstrap, clip_rho=1.0, clip_pg_rho=1.0):
    """vtrace.py:99-137 op for op in torch (the reference's per-t Python loop kept)"""
    import torch
    rhos = torch.exp(tlp - blp)                                                     # :99-101
    clipped_rhos = torch.clamp(rhos, max=clip_rho)                                  # :102-105
    cs = torch.clamp(rhos, max=1.0)                                                 # :107
    values_t_plus_1 = torch.cat([values[1:], bootstrap[None]], 0)                   # :110-111
    deltas = clipped_rhos * (rewards + discounts * values_t_plus_1 - values)        # :112-114
    acc = torch.zeros_like(bootstrap)                                               # :116
    out = []
    for t in reversed(range(values.shape[0])):                                      # :118-122
        acc = deltas[t] + discounts[t] * cs[t] * acc
        out.append(acc)
    vs = torch.stack(out[::-1]) + values                                            # :122-125
    vs_t_plus_1 = torch.cat([vs[1:], bootstrap[None]], 0)                           # :128-129
    pg_rhos = torch.clamp(rhos, max=clip_pg_rho)                                    # :131-134
    return vs, pg_rhos * (rewards + discounts * vs_t_plus_1 - values)               # :135-137


def time_vtrace_torch(threads=8):
    import torch
    n0 = torch.get_num_threads()
    torch.set_num_threads(threads)
    out = {}
    try:
        g = torch.Generator().manual_seed(0)
        for Tq, B, runs in ((49, 1024, 20), (127, 65536, 5)):
            x = [torch.randn(Tq, B, generator=g) for _ in range(5)]
            x[2] = torch.full((Tq, B), 0.99)
            boot = torch.randn(B, generator=g)
            for _ in range(3):
                vtrace_torch(x[0], x[1], x[2], x[3], x[4], boot)
            ts = []
            for _ in range(runs):
                t0 = time.perf_counter()
                vtrace_torch(x[0], x[1], x[2], x[3], x[4], boot)
                ts.append(time.perf_counter() - t0)
            s = float(np.median(ts))
            out['vtrace_torch_cpu_T%d_B%d' % (Tq, B)] = {'seconds': s, 'elements_per_s': Tq * B / s,
                                                         'GBps': Tq * B * 28 / s / 1e9, 'kind': 'port', 'cores': threads,
                                                         'what': 'vtrace.py:99-137 per-t loop as torch-CPU ops'}
    finally:
        torch.set_num_threads(n0)
    return out


def time_all(threads=8):
    out = {'a2c_learn_reference_torch_cpu': time_reference_a2c_learn(threads=threads)}
    out.update(time_vtrace_torch(threads=threads))
    return out


if __name__ == '__main__':
    import json
    print(json.dumps(time_all(), indent=1))
