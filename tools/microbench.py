"""Per-kernel timing of the scan kernels with HIP events (dev tool; bench.py is the contract).
Usage: python tools/microbench.py [--json out.json]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parl_amd import ops  # noqa: E402


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e-3  # median seconds


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--json', default=None)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    res = {}

    def vt(T, B):
        x = [torch.randn((T, B), device=dev) for _ in range(5)]
        boot = torch.randn(B, device=dev)
        s = timeit(lambda: ops.vtrace(x[0], x[1], x[2], x[3], x[4], boot))
        by = T * B * 28 + 4 * B
        res['vtrace_T%d_B%d' % (T, B)] = {'us': s * 1e6, 'GBps': by / s / 1e9, 'bytes': by}

    vt(49, 1024)
    vt(49, 20)
    vt(127, 262144)
    vt(127, 1 << 20)
    vt(49, 65536)

    def vl(T, B, A, tm):
        shp = (T, B) if tm else (B, T)
        bl = torch.randn(shp + (A, ), device=dev)
        tl = torch.randn(shp + (A, ), device=dev)
        act = torch.randint(0, A, shp, device=dev)
        rew = torch.randn(shp, device=dev)
        d = torch.rand(shp, device=dev) < 0.01
        val = torch.randn(shp, device=dev)
        s = timeit(lambda: ops.vtrace_from_logits(bl, tl, act, rew, d, val, 0.99, time_major=tm))
        by = T * B * (2 * A * 4 + 8 + 4 + 1 + 4 + 8)
        res['vtrace_logits_%s_T%d_B%d_A%d' % ('tm' if tm else 'em', T, B, A)] = {'us': s * 1e6, 'GBps': by / s / 1e9, 'bytes': by}

    vl(50, 1024, 6, True)
    vl(50, 1024, 6, False)
    vl(50, 65536, 6, True)
    vl(50, 65536, 6, False)
    vl(128, 262144, 6, True)

    def ga(T, B, f32):
        rew = torch.randn((T, B), device=dev)
        val = torch.randn((T, B), device=dev)
        d = (torch.rand((T, B), device=dev) < 0.01)
        d = d.float() if f32 else d
        nv = torch.randn(B, device=dev)
        ld = torch.zeros(B, device=dev) if f32 else None
        s = timeit(lambda: ops.gae(rew, val, d, nv, 0.99, 0.95, last_done=ld, done_convention=1 if f32 else 0))
        by = T * B * (20 if f32 else 17)
        res['gae_T%d_B%d_%s' % (T, B, 'f32' if f32 else 'u8')] = {'us': s * 1e6, 'GBps': by / s / 1e9, 'bytes': by}

    ga(20, 256, False)
    ga(2048, 4096, True)
    ga(128, 262144, False)

    adv = torch.randn(4096 * 2048, device=dev)
    idx = torch.randperm(4096 * 2048, device=dev)[:262144]
    s = timeit(lambda: ops.adv_normalize(adv, idx))
    res['adv_normalize_gather_262144'] = {'us': s * 1e6}
    s = timeit(lambda: ops.adv_normalize(adv))
    res['adv_normalize_8M'] = {'us': s * 1e6, 'GBps': adv.numel() * 12 / s / 1e9}
    lg = torch.randn((1024, 6), device=dev)
    s = timeit(lambda: ops.policy_sample(lg, 1, 2, 3))
    res['policy_sample_1024x6'] = {'us': s * 1e6}
    for n in (1024, 8192):
        obs = torch.randint(0, 256, (n, 4, 42, 42), dtype=torch.uint8, device=dev)
        w1, b1 = torch.randn(16, 4, 4, 4, device=dev), torch.randn(16, device=dev)
        w2, b2 = torch.randn(32, 16, 4, 4, device=dev), torch.randn(32, device=dev)
        out = torch.empty((n, 3872), device=dev)
        s = timeit(lambda: ops.atari42_conv12(obs, w1, b1, w2, b2, out=out))
        fl = n * 2.0 * (441 * 16 * 64 + 121 * 32 * 256)
        res['conv12_u8_mfma_n%d' % n] = {'us': s * 1e6, 'TFLOPs': fl / s / 1e12, 'GBps': n * (7056 + 15488) / s / 1e9}
    T, B, A = 50, 1024, 6
    bl, tl = torch.randn((T, B, A), device=dev), torch.randn((T, B, A), device=dev)
    act = torch.randint(0, A, (T, B), device=dev)
    rw, dn, vl = torch.randn((T, B), device=dev), torch.rand((T, B), device=dev) < 0.01, torch.randn((T, B), device=dev)
    s = timeit(lambda: ops.impala_loss(bl, tl, act, rw, dn, vl, 0.99))
    by = T * B * (2 * A * 4 + 8 + 4 + 1 + 4) + (T - 1) * B * 8 + T * B * (4 * A + 4)
    res['impala_loss_T50_B1024_A6 (incl. 5 allocations + zero fill)'] = {'us': s * 1e6, 'GBps': by / s / 1e9, 'bytes': by}
    from parl_amd.env import DeviceVectorEnv
    for dim in (42, 84):
        env = DeviceVectorEnv('PongNoFrameskip-v4', 1024, dim=dim, horizon=8, seed=1, device=dev)
        env.reset()
        env.step_async(torch.zeros(1024, dtype=torch.int64, device=dev))
        s = timeit(lambda: env._frame_post(1))
        by = 1024 * (2 * 33600 + dim * dim)
        res['frame_post_since_E1024_d%d' % dim] = {'us': s * 1e6, 'GBps': by / s / 1e9, 'bytes': by}
    for k, v in res.items():
        print(k, {a: (round(b, 2) if isinstance(b, float) else b) for a, b in v.items()})
    if args.json:
        json.dump(res, open(args.json, 'w'), indent=1)


if __name__ == '__main__':
    main()
