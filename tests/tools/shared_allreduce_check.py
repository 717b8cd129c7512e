"""two (or more) ranks on ONE GPU: parl_amd.dist.FlatGradAllReduce over the shared-device path (HIP IPC slots +
gloo barrier) gives every rank the SUM of the ranks' gradients, bit-identical on all ranks, call after call.
Launched by tests/test_gpu_dist.py with RANK / WORLD_SIZE / MASTER_* set."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from parl_amd import dist as pdist  # noqa: E402

rank, local, world = pdist.init(backend='gloo')
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
model = torch.nn.Sequential(torch.nn.Linear(300, 200), torch.nn.Linear(200, 7)).to(dev)
hook = pdist.FlatGradAllReduce(model)
n = hook.flat.numel()
for it in range(6):
    hook.zero_grad()
    g = torch.Generator(device=dev).manual_seed(1000 * it + rank)
    mine = torch.randn(n, device=dev, generator=g)
    hook.flat.copy_(mine)
    hook(model)
    expect = torch.zeros(n, device=dev)
    for r in range(world):   # rank order: the order the shared path adds in
        expect += torch.randn(n, device=dev, generator=torch.Generator(device=dev).manual_seed(1000 * it + r))
    assert torch.equal(hook.flat, expect), (it, float((hook.flat - expect).abs().max()))
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(hook.params, hook.views))
assert isinstance(hook._shared, pdist.SharedDeviceAllReduce), hook._shared
pdist.barrier()
print('SHARED_ALLREDUCE_OK rank %d of %d' % (rank, world), flush=True)
