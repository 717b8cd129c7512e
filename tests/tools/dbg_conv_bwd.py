import sys, os, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from parl_amd import ops
dev = torch.device('cuda:0')
def ref_grads(obs, w1, b1, w2, b2, dy):
    p = [t.double().clone().requires_grad_(True) for t in (w1, b1, w2, b2)]
    x = obs.double() / 255.0
    x = F.relu(F.conv2d(x, p[0], p[1], stride=2, padding=1))
    x = F.relu(F.conv2d(x, p[2], p[3], stride=2, padding=2)).flatten(1)
    (x * dy.double()).sum().backward()
    return [t.grad for t in p]
n = 5
g = torch.Generator().manual_seed(100 + n)
obs = torch.randint(0, 256, (n, 4, 42, 42), generator=g, dtype=torch.uint8)
obs[0, :, :5] = 0
w1 = torch.randn(16, 4, 4, 4, generator=g) * 0.2; b1 = torch.randn(16, generator=g) * 0.1
w2 = torch.randn(32, 16, 4, 4, generator=g) * 0.1; b2 = torch.randn(32, generator=g) * 0.1
dy = torch.randn(n, 3872, generator=g)
D = lambda t: t.to(dev)
a2 = ops.atari42_conv12(D(obs), D(w1), D(b1), D(w2), D(b2))
full = ops.atari42_conv12_backward(D(obs), D(w1), D(b1), D(w2), a2, D(dy))
names = ('dw1', 'db1', 'dw2', 'db2')
for i in range(n):
    one = ops.atari42_conv12_backward(D(obs[i:i+1]), D(w1), D(b1), D(w2), a2[i:i+1].contiguous(), D(dy[i:i+1]))
    r = ref_grads(obs[i:i+1], w1, b1, w2, b2, dy[i:i+1])
    print('obs', i, [(nm, float((a.cpu().double() - b).abs().max()), float(b.abs().max())) for nm, a, b in zip(names, one, r)])
    if i == 0: acc = [x.clone() for x in one]
    else: acc = [x + y for x, y in zip(acc, one)]
r = ref_grads(obs, w1, b1, w2, b2, dy)
print('sum of singles vs ref', [(nm, float((a.cpu().double() - b).abs().max())) for nm, a, b in zip(names, acc, r)])
print('full call vs ref     ', [(nm, float((a.cpu().double() - b).abs().max())) for nm, a, b in zip(names, full, r)])
# locate the error for the worst single
for i in range(n):
    one = ops.atari42_conv12_backward(D(obs[i:i+1]), D(w1), D(b1), D(w2), a2[i:i+1].contiguous(), D(dy[i:i+1]))
    r = ref_grads(obs[i:i+1], w1, b1, w2, b2, dy[i:i+1])
    e = (one[0].cpu().double() - r[0]).abs()
    if e.max() > 1e-3:
        idx = torch.nonzero(e > 1e-3)
        print('obs', i, 'dw1 bad entries', idx.shape[0], idx[:10].tolist())
        # check the a1 relu ties
        x = obs[i:i+1].double()/255
        z1 = F.conv2d(x, w1.double(), b1.double(), stride=2, padding=1)
        print(' min |z1|', float(z1.abs().min()))
        z1f = F.conv2d(obs[i:i+1].float()/255, w1, b1, stride=2, padding=1)
        print(' sign mismatches f32 vs f64', int(((z1f > 0) != (z1 > 0)).sum()))
