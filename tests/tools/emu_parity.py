"""Dev tool (GPU box): step the device env and the CPU oracle env side by side and report the
first divergence in detail."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import c_oracle  # noqa: E402
from parl_amd.env import DeviceVectorEnv, find_rom, GAMES  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--game', default='PongNoFrameskip-v4')
    ap.add_argument('--envs', type=int, default=4)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--dim', type=int, default=84)
    ap.add_argument('--max-episode-steps', type=int, default=400000)
    ap.add_argument('--no-cache', action='store_true')
    ap.add_argument('--seed', type=int, default=3)
    args = ap.parse_args()
    name = GAMES[args.game][0]
    E = args.envs
    t0 = time.time()
    dev = DeviceVectorEnv(args.game, E, dim=args.dim, horizon=8, seed=args.seed, max_episode_steps=args.max_episode_steps,
                          use_reset_cache=not args.no_cache)
    orc = c_oracle.VecEnv(find_rom(name), name, E, args.dim, seed=args.seed, max_episode_steps=args.max_episode_steps)
    dobs = dev.reset().cpu().numpy()
    torch.cuda.synchronize()
    print('device reset %.2fs jam=%x' % (time.time() - t0, int(dev.jam.item())))
    oobs = orc.reset()
    state_b = int(dev.states.numel() // E)

    def ram_dev(e):
        return dev.states.view(E, state_b)[e, :128].cpu().numpy()

    def report(tag, i):
        print('MISMATCH', tag, 'at step', i)
        for e in range(E):
            rd, ro = ram_dev(e), orc.ram(e)
            nz = np.nonzero(rd != ro)[0]
            print(' env', e, 'ram diff idx', nz[:16], 'dev', rd[nz[:8]], 'orc', ro[nz[:8]])
            fd = dev.raw_frames[e].cpu().numpy()
            fo = orc.raw_frames(e)
            d = np.argwhere(fd != fo)
            print('   raw frame diffs', len(d), d[:5])
        sys.exit(1)

    if not np.array_equal(dobs, oobs):
        report('reset obs', -1)
    rng = np.random.default_rng(0)
    A = dev.act_dim
    nd = 0
    for i in range(args.steps):
        a = rng.integers(0, A, E)
        o, r, d, info = dev.step(torch.from_numpy(a).to(dev.device))
        o, r, d = o.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy()
        oo, orr, od = orc.step(a)
        nd += int(od.sum())
        if not np.array_equal(r, orr):
            print('rewards', r, orr)
            report('reward', i)
        if not np.array_equal(d.astype(np.uint8), od):
            print('dones', d, od)
            report('done', i)
        if not np.array_equal(o, oo):
            report('obs', i)
        for e in range(E):
            if not np.array_equal(ram_dev(e), orc.ram(e)):
                report('ram', i)
        ln = info['episode_lengths'].cpu().numpy()
        for e in range(E):
            eps = orc.pop_episodes(e)
            if eps:
                assert ln[e] == eps[-1][1] and info['episode_returns'].cpu().numpy()[e] == eps[-1][0], (ln[e], eps)
            else:
                assert ln[e] == 0
    print('OK: %d steps x %d envs identical (dones seen: %d), jam=%x' % (args.steps, E, nd, int(dev.jam.item())))


if __name__ == '__main__':
    main()
