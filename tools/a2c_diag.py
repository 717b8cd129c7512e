"""Dev tool (GPU box): is A2C.learn on the 84x84 model doing its job at the example's batch (256 envs x 20 steps)?
(1) the gradients of one learn-shaped loss on REAL observations through the u8 kernel path against the float
(GEMM-lowered) path of the same weights; (2) 40 updates on one fixed batch through each path: vf_loss must fall;
(3) after those updates, the actors' no-grad forward against the learner's forward on the same weights."""
import copy
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import parl_amd as parl  # noqa: E402
from parl_amd.env import DeviceVectorEnv  # noqa: E402
from parl_amd.models import AtariModel84  # noqa: E402
from parl_amd.rollout import DeviceA2CRollout  # noqa: E402


def loss_of(model, obs, actions, adv, target):
    logits, values = model.policy_and_value(obs)
    lp = F.log_softmax(logits, dim=1)
    pi = -(lp.gather(1, actions.unsqueeze(1)).squeeze(1) * adv).sum()
    vf = 0.5 * (values - target).square().sum()
    ent = -(lp.exp() * lp).sum()
    return pi + 0.5 * vf - 0.01 * ent, vf


def main():
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    T = 20
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    env = DeviceVectorEnv('PongNoFrameskip-v4', E, dim=84, horizon=T, seed=0, device=dev)
    model = AtariModel84(env.act_dim).to(dev)
    ro = DeviceA2CRollout(env, T, 0.99, 1.0, seed=1000)
    for _ in range(12):
        batch = ro.collect(model)
    torch.cuda.synchronize()
    obs, actions = batch['obs'].clone(), batch['actions'].clone()
    adv, target = batch['advantages'].clone(), batch['target_values'].clone()
    print('rows', obs.shape[0], 'obs mean', float(obs.float().mean()), 'adv |max|', float(adv.abs().max()),
          'target |max|', float(target.abs().max()), 'finite', bool(torch.isfinite(adv).all() and torch.isfinite(target).all()))
    # (1) gradients: kernel path vs float path
    mk, mf = copy.deepcopy(model), copy.deepcopy(model)
    lk, vk = loss_of(mk, obs, actions, adv, target)
    lk.backward()
    lf, vf = loss_of(mf, obs.float(), actions, adv, target)
    lf.backward()
    print('loss kernel %.6f float %.6f   vf %.6f %.6f' % (float(lk), float(lf), float(vk), float(vf)))
    for (name, pk), pf in zip(mk.named_parameters(), mf.parameters()):
        gk, gf = pk.grad, pf.grad
        print('  %-18s |g| kernel %.5e float %.5e  max|diff| %.3e  finite %s' %
              (name, float(gk.norm()), float(gf.norm()), float((gk - gf).abs().max()), bool(torch.isfinite(gk).all())))
    # (2) 40 updates on the fixed batch
    for tag, m, o in (('kernel', copy.deepcopy(model), obs), ('float', copy.deepcopy(model), obs.float())):
        alg = parl.algorithms.A2C(m, vf_loss_coeff=0.5)
        traj, norms = [], []
        for it in range(40):
            out = alg.learn(o, actions, adv, target, 1e-3, -0.01)
            traj.append(float(out[2]))
        print(tag, 'vf_loss over 40 updates:', ' '.join('%.1f' % v for v in traj[::4]), ' last %.2f' % traj[-1])
        if tag == 'kernel':
            with torch.no_grad():
                la, va = m.policy_and_value(obs[:512])          # the actors' kernels (cached weight layouts)
            with torch.enable_grad():
                ll, vl = m.policy_and_value(obs[:512])          # the learner's forward
            print('  actor vs learner forward after the updates: max|dlogit| %.3e  max|dvalue| %.3e' %
                  (float((la - ll).abs().max()), float((va - vl).abs().max())))
    # (3) the rollout object itself across updates: values it records against a fresh forward
    alg = parl.algorithms.A2C(model, vf_loss_coeff=0.5)
    for it in range(6):
        b = ro.collect(model)
        out = alg.learn(b['obs'], b['actions'], b['advantages'], b['target_values'], 1e-3, -0.01)
        with torch.enable_grad():
            _, vl = model.policy_and_value(b['obs'][:E])
        print('  update %d: vf_loss %.2f  total grad norm %.4e' %
              (it, float(out[2]), float(torch.nn.utils.clip_grad_norm_(model.parameters(), 1e30))))


if __name__ == '__main__':
    main()
