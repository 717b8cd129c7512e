"""Dev tool (GPU box): A2C.learn of the 84x84 model at 5,120 rows (configs[1]) and IMPALA.learn of the 42x42 model
at 51,200 rows, 12 times each — run it under `rocprofv3 --kernel-trace --stats` (tools/learn84_prof.sh) for the
per-kernel table of one learner update.  PARL_HIP_LIB selects the library."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import parl_amd as parl  # noqa: E402
from parl_amd.models import AtariModel42, AtariModel84  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(0)
which = sys.argv[1] if len(sys.argv) > 1 else '84'
if which == '84':
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 5120
    m = AtariModel84(6).to(dev)
    alg = parl.algorithms.A2C(m, vf_loss_coeff=0.5)
    o = torch.randint(0, 256, (rows, 4, 84, 84), dtype=torch.uint8, device=dev)
    a = torch.randint(0, 6, (rows, ), device=dev)
    adv, tgt = torch.randn(rows, device=dev), torch.randn(rows, device=dev)
    fn = lambda: alg.learn(o, a, adv, tgt, 1e-4, -0.01)  # noqa: E731
else:
    T, E = 50, 1024
    m = AtariModel42(6).to(dev)
    alg = parl.algorithms.IMPALA(m, sample_batch_steps=T, gamma=0.99, vf_loss_coeff=0.5, clip_rho_threshold=1.0,
                                 clip_pg_rho_threshold=1.0)
    o = torch.randint(0, 256, (T * E, 4, 42, 42), dtype=torch.uint8, device=dev)
    act = torch.randint(0, 6, (T * E, ), device=dev)
    bl, rew = torch.randn((T * E, 6), device=dev), torch.randn(T * E, device=dev)
    dn = torch.rand(T * E, device=dev) < 0.01
    fn = lambda: alg.learn(o, act, bl, rew, dn, 1e-4, -0.01, time_major=True)  # noqa: E731
for _ in range(12):
    fn()
torch.cuda.synchronize()
