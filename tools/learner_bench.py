"""Learner-side timing (dev tool): the fused conv1+conv2 forward / backward kernels at the learner
batch of the bench (51,200 observations) and one whole IMPALA.learn update, HIP-event timed.
Usage: python tools/learner_bench.py [--n 51200] [--json out.json]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import parl_amd as parl  # noqa: E402
from parl_amd import ops  # noqa: E402
from parl_amd.models import AtariModel42  # noqa: E402
from microbench import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=51200)
    ap.add_argument('--json', default=None)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    n = args.n
    res = {}
    obs = torch.randint(0, 256, (n, 4, 42, 42), dtype=torch.uint8, device=dev)
    w1, b1 = torch.randn(16, 4, 4, 4, device=dev) * 0.2, torch.randn(16, device=dev) * 0.1
    w2, b2 = torch.randn(32, 16, 4, 4, device=dev) * 0.1, torch.randn(32, device=dev) * 0.1
    out = torch.empty((n, 3872), device=dev)
    s = timeit(lambda: ops.atari42_conv12(obs, w1, b1, w2, b2, out=out), iters=10)
    fl = n * 2.0 * (441 * 16 * 64 + 121 * 32 * 256)
    res['conv12_fwd_n%d' % n] = {'ms': s * 1e3, 'TFLOPs': fl / s / 1e12}
    dy = torch.randn((n, 3872), device=dev)
    s = timeit(lambda: ops.atari42_conv12_backward(obs, w1, b1, w2, out, dy), iters=10)
    flb = n * 2048.0 * (448 + 992 + 928 + 444)
    res['conv12_bwd_n%d' % n] = {'ms': s * 1e3, 'TFLOPs_issued': flb / s / 1e12,
                                 'GBps': n * (7056 + 2 * 15488) / s / 1e9}
    T = 50
    E = n // T
    torch.manual_seed(0)
    model = AtariModel42(6).to(dev)
    alg = parl.algorithms.IMPALA(model, sample_batch_steps=T, gamma=0.99, vf_loss_coeff=0.5,
                                 clip_rho_threshold=1.0, clip_pg_rho_threshold=1.0)
    ob = obs[:T * E]
    act = torch.randint(0, 6, (T * E, ), device=dev)
    bl = torch.randn((T * E, 6), device=dev)
    rew = torch.randn(T * E, device=dev)
    dn = torch.rand(T * E, device=dev) < 0.01
    for name, o in (('learn_u8_fused_conv', ob), ('learn_f32_gemm_conv', None)):
        if o is None:
            o = ob.float()
        s = timeit(lambda: alg.learn(o, act, bl, rew, dn, 1e-4, -0.01, time_major=True), iters=5, warmup=2)
        res['%s_T%d_E%d' % (name, T, E)] = {'ms': s * 1e3}
    # the 84x84 model (A2C network): one A2C.learn at the configs[1] batch and at an IMPALA-sized one
    from parl_amd.models import AtariModel84
    m84 = AtariModel84(6).to(dev)
    a2c = parl.algorithms.A2C(m84, vf_loss_coeff=0.5)
    for rows in (5120, 25600):
        o84 = torch.randint(0, 256, (rows, 4, 84, 84), dtype=torch.uint8, device=dev)
        a84 = torch.randint(0, 6, (rows, ), device=dev)
        adv, tgt = torch.randn(rows, device=dev), torch.randn(rows, device=dev)
        s = timeit(lambda: a2c.learn(o84, a84, adv, tgt, 1e-4, -0.01), iters=5, warmup=2)
        res['a2c_learn_84_u8_mfma_rows%d' % rows] = {'ms': s * 1e3}
        if rows == 5120:
            of = o84.float()
            s = timeit(lambda: a2c.learn(of, a84, adv, tgt, 1e-4, -0.01), iters=5, warmup=2)
            res['a2c_learn_84_f32_gemm_rows%d' % rows] = {'ms': s * 1e3}
        del o84
    for m in (1024, 8192):
        a1 = torch.relu(torch.randn(m, 32, 20, 20, device=dev))
        w2c, b2c = torch.randn(64, 32, 4, 4, device=dev) * 0.05, torch.zeros(64, device=dev)
        w3c, b3c = torch.randn(64, 64, 3, 3, device=dev) * 0.05, torch.zeros(64, device=dev)
        s = timeit(lambda: ops.atari84_conv23(a1, w2c, b2c, w3c, b3c), iters=10)
        res['conv23_84_fwd_n%d' % m] = {'ms': s * 1e3, 'TFLOPs': m * 2.0 * (121 * 512 * 64 + 81 * 576 * 64) / s / 1e12}
    for k, v in res.items():
        print(k, {a: round(b, 3) for a, b in v.items()})
    if args.json:
        json.dump(res, open(args.json, 'w'), indent=1)


if __name__ == '__main__':
    main()
