"""Dev tool (GPU box): where the ACTOR stream's time goes inside the pipeline.  The headline pipeline (1024 Pong envs,
T = 50, 51 hipGraph updates of 1000 rows per rollout) with the rollout enqueued EAGERLY (PARL_AMD_ROLLOUT_GRAPH=0) and
three HIP events per env step on the actor stream: before the policy trunk (conv12 + trunk GEMM), after it, after the
env launch.  Per step: chain = trunk, env = the env launch; events are recorded back to back, so a kernel that cannot
start (no free CU slots: the learner's workgroups) shows up INSIDE its interval.  The same rollout is then timed with
the learner stream idle.  Usage: python tools/pipeline_timeline.py"""
import os
import sys

os.environ['PARL_AMD_ROLLOUT_GRAPH'] = '0'
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import parl_amd as parl  # noqa: E402
from parl_amd.env import DeviceVectorEnv  # noqa: E402
from parl_amd.models import AtariModel42  # noqa: E402
from parl_amd.rollout import AsyncActorLearner  # noqa: E402

dev = torch.device('cuda:0')
E, T = 1024, 50
torch.manual_seed(0)
env = DeviceVectorEnv('PongNoFrameskip-v4', E, dim=42, horizon=T, seed=1234, device=dev)
model = AtariModel42(env.act_dim).to(dev)
alg = parl.algorithms.IMPALA(model, sample_batch_steps=T, gamma=0.99, vf_loss_coeff=0.5, clip_rho_threshold=1.0,
                             clip_pg_rho_threshold=1.0)
pipe = AsyncActorLearner(alg, [env], T, seed=99, train_batch_size=1000)
lr_s = parl.utils.PiecewiseScheduler([(0, 0.001), (20000, 0.0005), (40000, 0.0001)])
ent_s = parl.utils.PiecewiseScheduler([(0, -0.01)])
ro = pipe.rollout
marks = []
record = [False]
orig_hidden, orig_step = pipe.actor_model.policy_hidden, env.step_policy_async


def hidden(obs):
    if record[0]:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append(('a', e))
    h = orig_hidden(obs)
    if record[0]:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append(('b', e))
    return h


def step_policy(*a, **k):
    r = orig_step(*a, **k)
    if record[0]:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append(('c', e))
    return r


pipe.actor_model.policy_hidden = hidden
env.step_policy_async = step_policy


def summarise(tag):
    torch.cuda.synchronize()
    chain, envt, rolls = [], [], []
    i = 0
    first = None
    while i + 2 < len(marks) + 1 and i + 2 < len(marks):
        a, b, c = marks[i][1], marks[i + 1][1], marks[i + 2][1]
        chain.append(a.elapsed_time(b) * 1e3)
        envt.append(b.elapsed_time(c) * 1e3)
        if (i // 3) % T == 0:
            first = a
        if (i // 3) % T == T - 1:
            rolls.append(first.elapsed_time(c))
        i += 3
    chain.sort()
    envt.sort()
    med = lambda x: x[len(x) // 2]  # noqa: E731
    print('%-34s rollout %.2f ms | per step: trunk (conv12 + GEMM) median %.1f us (p10 %.1f, p90 %.1f) | env launch median %.1f us '
          '(p10 %.1f, p90 %.1f) | sum of medians x %d = %.2f ms' %
          (tag, sum(rolls) / max(len(rolls), 1), med(chain), chain[len(chain) // 10], chain[len(chain) * 9 // 10], med(envt),
           envt[len(envt) // 10], envt[len(envt) * 9 // 10], T, (med(chain) + med(envt)) * T * 1e-3), flush=True)
    del marks[:]


pipe.prime()
for _ in range(4):
    pipe.step(lr_s, ent_s)
pipe.synchronize()
record[0] = True
import time  # noqa: E402
t0 = time.time()
K = 12
for _ in range(K):
    pipe.step(lr_s, ent_s)
pipe.synchronize()
dt = time.time() - t0
print('pipeline (eager rollout): %.2f ms per step = %.2f M frames/s' % (dt / K * 1e3, K * T * E * 4 / dt / 1e6))
summarise('beside the learner (51 updates):')
# the same rollouts with the learner stream idle
with torch.cuda.stream(pipe.actor_stream):
    for _ in range(3):
        ro.collect_begin()
        ro.collect_steps(pipe.actor_model)
        ro.collect_end()
    pipe.synchronize()
    del marks[:]
    for _ in range(6):
        ro.collect_begin()
        ro.collect_steps(pipe.actor_model)
        ro.collect_end()
pipe.synchronize()
summarise('alone (learner stream idle):')
