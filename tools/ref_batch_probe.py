"""Dev tool: one IMPALA.learn at the reference's learner batch (train_batch_size = 1000 rows = 20 sequences x
T = 50), eager vs hipGraph replay (parl_amd.algorithms.impala.graphed.GraphedLearn), HIP-event + wall timed."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import parl_amd as parl  # noqa: E402
from parl_amd.algorithms.impala.graphed import GraphedLearn  # noqa: E402
from parl_amd.models import AtariModel42, AtariModel84  # noqa: E402


def main():
    dim = int(os.environ.get('DIM', '42'))
    B = int(os.environ.get('B', '20'))
    dev = torch.device('cuda:0')
    T, A = 50, 6
    torch.manual_seed(0)
    model = (AtariModel42 if dim == 42 else AtariModel84)(A).to(dev)
    with torch.no_grad():
        model.policy_fc.weight.mul_(0.05)
        model.value_fc.weight.mul_(0.05)
    alg = parl.algorithms.IMPALA(model, sample_batch_steps=T, gamma=0.99, vf_loss_coeff=0.5, clip_rho_threshold=1.0,
                                 clip_pg_rho_threshold=1.0)
    N = T * B
    batch = {'obs': torch.randint(0, 256, (N, 4, dim, dim), dtype=torch.uint8, device=dev),
             'actions': torch.randint(0, A, (N, ), device=dev),
             'behaviour_logits': torch.randn((N, A), device=dev),
             'rewards': torch.randn(N, device=dev),
             'dones': torch.rand(N, device=dev) < 0.01}

    def eager():
        alg.learn(batch['obs'], batch['actions'], batch['behaviour_logits'], batch['rewards'], batch['dones'], 1e-3,
                  -0.01, time_major=True)

    def timeit(fn, iters=50):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(iters):
            fn()
        t1 = time.time()
        torch.cuda.synchronize()
        t2 = time.time()
        return (t1 - t0) / iters * 1e3, (t2 - t0) / iters * 1e3

    print('eager  ms/update (host enqueue, to completion): %.3f %.3f' % timeit(eager))
    gl = GraphedLearn(alg, B, (4, dim, dim), A)

    def graphed():
        gl.load(batch, 0, B)
        gl.replay(1e-3)

    print('graph  ms/update (host enqueue, to completion): %.3f %.3f' % timeit(graphed))
    print('parameter update inside the graph:', 'ops.ClipAdam (2 launches)' if gl._clip_adam else 'clip_grad_norm_ + torch fused Adam')

    def replay_only():
        gl.replay(1e-3)

    print('replay ms/update (host enqueue, to completion): %.3f %.3f' % timeit(replay_only))
    print('stats', gl.pop_stats())


if __name__ == '__main__':
    main()
