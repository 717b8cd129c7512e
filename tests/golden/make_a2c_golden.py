"""a2c_learn.npz: the reference's own torch A2C.learn (parl/algorithms/torch/a2c.py:40-81) and its
own torch ActorCritic model (benchmark/torch/a2c/atari_model.py:23-104), imported from
/root/reference with the 5-stub shim of SURVEY.md A4 and run on CPU: initial state_dict, two
batches (uint8 observations, actions, advantages, target values), the four losses of each
learn() call, every parameter's GRADIENT as Adam consumed it (captured at optimizer.step(), i.e.
after clip_grad_norm_(…, 40.0), a2c.py:66-68), the parameters after the two updates,
and prob_and_value / predict outputs.
To keep the fixture small the initial parameters are drawn from a seeded numpy generator
(`init_weights`, xavier-normal scale like the reference's _init_parameters) and loaded into the
reference model — the test regenerates them — and the 2.65 M-element fc weight is stored after
the updates as a strided sample plus its sum and L2 norm; every other parameter is stored whole.
Build-container only.

    python tests/golden/make_a2c_golden.py
"""
import importlib.util
import os
import sys

sys.dont_write_bytecode = True  # modules are imported from /root/reference by path: never leave a __pycache__ there

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_ppo_golden import REF, _import_reference_parl  # noqa: E402

SHAPES = [('conv1.weight', (32, 4, 8, 8)), ('conv1.bias', (32, )), ('conv2.weight', (64, 32, 4, 4)),
          ('conv2.bias', (64, )), ('conv3.weight', (64, 64, 3, 3)), ('conv3.bias', (64, )),
          ('fc.weight', (512, 5184)), ('fc.bias', (512, )), ('fc_pi.weight', (None, 512)), ('fc_pi.bias', (None, )),
          ('fc_v.weight', (1, 512)), ('fc_v.bias', (1, ))]
FC_STRIDE = 97


def init_weights(act_dim, seed=3):
    """deterministic xavier-normal-scaled parameters (numpy, so that any box regenerates them)"""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shp in SHAPES:
        shp = tuple(act_dim if d is None else d for d in shp)
        if name.endswith('bias'):
            out[name] = (0.01 * rng.standard_normal(shp)).astype(np.float32)
        else:
            rf = int(np.prod(shp[2:])) if len(shp) > 2 else 1
            std = (2.0 / (shp[1] * rf + shp[0] * rf)) ** 0.5
            out[name] = (std * rng.standard_normal(shp)).astype(np.float32)
    return out


if __name__ == '__main__':
    import torch
    parl, _ = _import_reference_parl()
    from parl.algorithms import A2C
    spec = importlib.util.spec_from_file_location('ref_a2c_atari_model',
                                                  os.path.join(REF, 'benchmark/torch/a2c/atari_model.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    torch.manual_seed(3)
    torch.set_num_threads(4)
    A, N = 6, 48
    model = mod.ActorCritic(A)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in init_weights(A).items()})
    alg = A2C(model, {'vf_loss_coeff': 0.5, 'learning_rate': 0.001})
    rng = np.random.default_rng(11)
    out = {'dims': np.array([A, N])}
    grads = {}
    opt_step = alg.optimizer.step

    def recording_step(*a, **kw):  # learn() zeroes the gradients right after the step (a2c.py:69)
        grads.update({k: prm.grad.detach().clone().numpy() for k, prm in model.named_parameters()})
        return opt_step(*a, **kw)

    alg.optimizer.step = recording_step
    for step, (lr, ec) in enumerate([(1e-3, -0.01), (7e-4, -0.02)]):
        # block-structured observations (like frames), not white noise
        obs = np.repeat(np.repeat(rng.integers(0, 256, (N, 4, 12, 12), dtype=np.uint8), 7, 2), 7, 3)
        act = rng.integers(0, A, N).astype(np.int64)
        adv = rng.standard_normal(N).astype(np.float32)
        tgt = rng.standard_normal(N).astype(np.float32)
        if step == 0:
            with torch.no_grad():
                p, v = alg.prob_and_value(torch.from_numpy(obs).float())
                out['probs0'], out['values0'] = p.numpy(), v.numpy()
                out['predict0'] = alg.predict(torch.from_numpy(obs).float()).numpy()
        losses = alg.learn(torch.from_numpy(obs).float(), torch.from_numpy(act), torch.from_numpy(adv),
                           torch.from_numpy(tgt), lr, ec)
        out['step%d/obs' % step], out['step%d/actions' % step] = obs, act
        out['step%d/advantages' % step], out['step%d/target_values' % step] = adv, tgt
        out['step%d/lr_ec' % step] = np.array([lr, ec])
        out['step%d/losses' % step] = np.array([float(x) for x in losses])
        for k, g in grads.items():
            if k == 'fc.weight':
                out['step%d/grad_sample/%s' % (step, k)] = g.reshape(-1)[::FC_STRIDE].copy()
                out['step%d/grad_stats/%s' % (step, k)] = np.array([np.abs(g).max(), np.sqrt((g.astype(np.float64) ** 2).sum())])
            else:
                out['step%d/grad/%s' % (step, k)] = g.copy()
    # a second, INDEPENDENT gradient check at regenerable weights (init_weights(seed=4)): the second update's
    # gradient above is taken at parameters that already went through one Adam step, whose sign-like first
    # step amplifies float32 rounding of near-zero gradients into +-lr — a test against it cannot be tight
    model2 = mod.ActorCritic(A)
    model2.load_state_dict({k: torch.from_numpy(v) for k, v in init_weights(A, seed=4).items()})
    alg2 = A2C(model2, {'vf_loss_coeff': 0.5, 'learning_rate': 0.001})
    grads2 = {}
    opt_step2 = alg2.optimizer.step

    def recording_step2(*a, **kw):
        grads2.update({k: prm.grad.detach().clone().numpy() for k, prm in model2.named_parameters()})
        return opt_step2(*a, **kw)

    alg2.optimizer.step = recording_step2
    obs = np.repeat(np.repeat(rng.integers(0, 256, (N, 4, 12, 12), dtype=np.uint8), 7, 2), 7, 3)
    act = rng.integers(0, A, N).astype(np.int64)
    adv = rng.standard_normal(N).astype(np.float32)
    tgt = rng.standard_normal(N).astype(np.float32)
    losses = alg2.learn(torch.from_numpy(obs).float(), torch.from_numpy(act), torch.from_numpy(adv),
                        torch.from_numpy(tgt), 5e-4, -0.01)
    out['indep/obs'], out['indep/actions'], out['indep/advantages'], out['indep/target_values'] = obs, act, adv, tgt
    out['indep/lr_ec'] = np.array([5e-4, -0.01])
    out['indep/losses'] = np.array([float(x) for x in losses])
    for k, g in grads2.items():
        if k == 'fc.weight':
            out['indep/grad_sample/' + k] = g.reshape(-1)[::FC_STRIDE].copy()
            out['indep/grad_stats/' + k] = np.array([np.abs(g).max(), np.sqrt((g.astype(np.float64) ** 2).sum())])
        else:
            out['indep/grad/' + k] = g.copy()
    for k, v in model.state_dict().items():
        w = v.detach().numpy()
        if k == 'fc.weight':
            out['final_sample/' + k] = w.reshape(-1)[::FC_STRIDE].copy()
            out['final_stats/' + k] = np.array([w.astype(np.float64).sum(), np.sqrt((w.astype(np.float64) ** 2).sum())])
        else:
            out['final/' + k] = w.copy()
    p = os.path.join(HERE, 'a2c_learn.npz')
    np.savez_compressed(p, **out)
    print(p, os.path.getsize(p), 'bytes; losses', out['step0/losses'], out['step1/losses'])
