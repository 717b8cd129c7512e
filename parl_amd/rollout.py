"""On-device IMPALA actor: the work of examples/IMPALA/actor.py:54-91 (Actor.sample) for E
envs at once, with no host round trip per step.

Reference flow per step (actor.py:58-76): agent.sample(obs) -> softmax probs to the host ->
np.random.choice per env (atari_agent.py:38-40) -> vector_env.step -> Python lists; after T steps
the lists are merged env-major and pickled to the learner (actor.py:78-89).
Here per step: policy forward (PyTorch-ROCm) -> ops.policy_sample (softmax + inverse-CDF draw,
Philox uniforms) -> DeviceVectorEnv.step_async (emulator + frame_post kernels) writing rewards /
dones straight into the [T,E] slabs of this buffer.  The batch handed to the learner is
TIME-major ([t0 all envs, t1 all envs, ...]); IMPALA.learn(time_major=True) consumes it without
a transpose (sum-reduced losses are order independent, impala.py:67-79)."""
import torch

from . import ops


class DeviceRollout(object):
    def __init__(self, env, sample_batch_steps, seed=0, n_buffers=1):
        """n_buffers > 1: successive collect() calls fill the trajectory slabs round-robin, so a
        learner on another stream can still read batch i-1 while batch i is being written
        (AsyncActorLearner below)."""
        assert env.horizon >= sample_batch_steps, 'env ring too short for the rollout'
        self.env, self.T, self.seed = env, int(sample_batch_steps), int(seed)
        E, A, dev = env.envs_num, env.act_dim, env.device
        T = self.T
        self._bufs = []
        for _ in range(int(n_buffers)):
            self._bufs.append({
                'actions': torch.zeros((T, E), dtype=torch.int64, device=dev),
                'behaviour_logits': torch.zeros((T, E, A), dtype=torch.float32, device=dev),
                'rewards': torch.zeros((T, E), dtype=torch.float32, device=dev),
                'dones': torch.zeros((T, E), dtype=torch.uint8, device=dev),
                'obs': torch.zeros((T * E, 4, env.dim, env.dim), dtype=torch.uint8, device=dev),
            })
        self._cur = -1
        self._select(0)
        self._obs_step = torch.zeros((E, 4, env.dim, env.dim), dtype=torch.uint8, device=dev)
        self._slots = (torch.arange(T, dtype=torch.int32, device=dev) + 3).repeat_interleave(E)
        self._envs = torch.arange(E, dtype=torch.int32, device=dev).repeat(T)
        self.step_count = 0  # Philox offset: one uniform per (global step, env)
        # MonitorEnv statistics (atari_wrappers.py:44-100), reduced on the device
        self.ep_count = torch.zeros((), dtype=torch.float64, device=dev)
        self.ep_return_sum = torch.zeros((), dtype=torch.float64, device=dev)
        self.ep_length_sum = torch.zeros((), dtype=torch.float64, device=dev)
        self.started = False

    def _select(self, k):
        b = self._bufs[k]
        self.actions, self.behaviour_logits = b['actions'], b['behaviour_logits']
        self.rewards, self.dones, self.obs = b['rewards'], b['dones'], b['obs']

    @torch.no_grad()
    def collect(self, model):
        """Run T env steps with `model` as behaviour policy; returns the time-major batch."""
        env = self.env
        self._cur = (self._cur + 1) % len(self._bufs)
        self._select(self._cur)
        if not self.started:
            env.reset()
            self.started = True
        else:
            env.roll()
        for t in range(self.T):
            obs = env.current_obs(self._obs_step)
            logits = model.policy(obs)
            self.behaviour_logits[t].copy_(logits)
            ops.policy_sample_into(logits, self.actions[t], self.seed, self.step_count, env.env_id0)
            env.step_async(self.actions[t], self.rewards[t], self.dones[t])
            closed = env.ep_lengths > 0
            self.ep_count += closed.sum()
            self.ep_return_sum += (env.ep_returns * closed).sum()
            self.ep_length_sum += (env.ep_lengths * closed).sum()
            self.step_count += 1
        env.gather(self._slots, self._envs, self.obs)
        E = env.envs_num
        return {
            'obs': self.obs,
            'actions': self.actions.reshape(self.T * E),
            'behaviour_logits': self.behaviour_logits.reshape(self.T * E, -1),
            'rewards': self.rewards.reshape(self.T * E),
            'dones': self.dones.reshape(self.T * E).bool(),
        }

    def pop_episode_stats(self):
        """(episodes closed, mean unclipped return, mean length in emulated frames); syncs."""
        n = float(self.ep_count.item())
        r = float(self.ep_return_sum.item())
        l = float(self.ep_length_sum.item())
        self.ep_count.zero_()
        self.ep_return_sum.zero_()
        self.ep_length_sum.zero_()
        return n, (r / n if n else None), (l / n if n else None)


class DeviceA2CRollout(object):
    """On-device A2C actor: the work of examples/A2C/actor.py:51-101 (Actor.sample) for E envs.

    Reference flow: per step agent.sample (probs + values to the host, np.random.choice per env) ->
    vector_env.step; whenever an env finishes an episode or the rollout ends, a batch-1 value
    forward for its next obs and calc_gae on that (env, segment) (actor.py:73-85).
    Here: policy_and_value forward -> ops.policy_sample -> DeviceVectorEnv.step_async for T steps,
    ONE batched value forward for the bootstrap of all envs, and ONE ops.gae launch over the
    [T,E] slabs whose done mask reproduces the per-segment semantics (next_value = 0 after a
    terminal step, carry reset).  Rows come out time-major; A2C's losses are sums (a2c.py:67-79),
    so the order is immaterial."""

    def __init__(self, env, sample_batch_steps, gamma, lam, seed=0):
        assert env.horizon >= sample_batch_steps, 'env ring too short for the rollout'
        self.env, self.T, self.seed = env, int(sample_batch_steps), int(seed)
        self.gamma, self.lam = float(gamma), float(lam)
        E, dev, T = env.envs_num, env.device, self.T
        self.actions = torch.zeros((T, E), dtype=torch.int64, device=dev)
        self.values = torch.zeros((T, E), dtype=torch.float32, device=dev)
        self.rewards = torch.zeros((T, E), dtype=torch.float32, device=dev)
        self.dones = torch.zeros((T, E), dtype=torch.uint8, device=dev)
        self.obs = torch.zeros((T * E, 4, env.dim, env.dim), dtype=torch.uint8, device=dev)
        self._obs_step = torch.zeros((E, 4, env.dim, env.dim), dtype=torch.uint8, device=dev)
        self._slots = (torch.arange(T, dtype=torch.int32, device=dev) + 3).repeat_interleave(E)
        self._envs = torch.arange(E, dtype=torch.int32, device=dev).repeat(T)
        self.step_count = 0
        self.ep_count = torch.zeros((), dtype=torch.float64, device=dev)
        self.ep_return_sum = torch.zeros((), dtype=torch.float64, device=dev)
        self.ep_length_sum = torch.zeros((), dtype=torch.float64, device=dev)
        self.started = False

    @torch.no_grad()
    def collect(self, model):
        env = self.env
        if not self.started:
            env.reset()
            self.started = True
        else:
            env.roll()
        for t in range(self.T):
            obs = env.current_obs(self._obs_step)
            logits, values = model.policy_and_value(obs)
            self.values[t].copy_(values)
            ops.policy_sample_into(logits, self.actions[t], self.seed, self.step_count, env.env_id0)
            env.step_async(self.actions[t], self.rewards[t], self.dones[t])
            closed = env.ep_lengths > 0
            self.ep_count += closed.sum()
            self.ep_return_sum += (env.ep_returns * closed).sum()
            self.ep_length_sum += (env.ep_lengths * closed).sum()
            self.step_count += 1
        next_value = model.value(env.current_obs(self._obs_step))  # ignored where the last step was terminal
        adv, target = ops.gae(self.rewards, self.values, self.dones, next_value, self.gamma, self.lam)
        env.gather(self._slots, self._envs, self.obs)
        n = self.T * env.envs_num
        return {'obs': self.obs, 'actions': self.actions.reshape(n), 'advantages': adv.reshape(n),
                'target_values': target.reshape(n)}

    pop_episode_stats = DeviceRollout.pop_episode_stats


class AsyncActorLearner(object):
    """IMPALA's actor / learner decoupling (examples/IMPALA/train.py:155-194: sample threads fill a
    queue while the learn thread drains it; actors act with parameters that lag the learner by up
    to `params_broadcast_interval` updates and V-trace corrects for the lag) on ONE GPU:
    two HIP streams instead of threads and processes.

      actor stream    weights snapshot -> T env steps (policy fwd, sample, emulator, frame_post) -> batch i
      learner stream  IMPALA.learn on batch i-1 (fwd, fused V-trace, bwd, [all-reduce], clip, Adam)

    Both are enqueued by one host thread; events order them.  The emulator kernel keeps one
    wavefront per SIMD busy and is latency-bound, so the learner's GEMMs run in the issue slots it
    leaves free.  The actor's parameter snapshot plays the role of the reference actor's
    `set_weights` (actor.py:103-104): it is refreshed from the learner before every rollout, so the
    behaviour policy lags the learner by exactly one update."""

    def __init__(self, alg, env, sample_batch_steps, seed=0):
        import copy
        self.alg, self.env = alg, env
        self.rollout = DeviceRollout(env, sample_batch_steps, seed=seed, n_buffers=2)
        self.actor_model = copy.deepcopy(alg.model)
        for p in self.actor_model.parameters():
            p.requires_grad_(False)
        dev = env.device
        # the rollout is the critical path (latency-bound emulator): high-priority queue for it
        self.actor_stream = torch.cuda.Stream(device=dev, priority=-1)
        self.learn_stream = torch.cuda.Stream(device=dev, priority=0)
        self.weights_ready = torch.cuda.Event()
        self.snapshot_done = torch.cuda.Event()
        self.batch_ready = [torch.cuda.Event(), torch.cuda.Event()]
        self.batch_free = [torch.cuda.Event(), torch.cuda.Event()]
        self.pending = None  # (batch, buffer index) collected but not yet learned
        self._src = [p for p in alg.model.parameters()] + [b for b in alg.model.buffers()]
        self._dst = [p for p in self.actor_model.parameters()] + [b for b in self.actor_model.buffers()]
        cur = torch.cuda.current_stream(dev)
        self.actor_stream.wait_stream(cur)  # env construction (reset cache, tables) ran on `cur`
        self.learn_stream.wait_stream(cur)
        self.weights_ready.record(cur)
        for e in self.batch_free:
            e.record(cur)

    def _snapshot(self):
        """actor parameters <- learner parameters (the reference actor's set_weights)"""
        with torch.cuda.stream(self.actor_stream):
            self.actor_stream.wait_event(self.weights_ready)
            with torch.no_grad():
                torch._foreach_copy_(self._dst, self._src)
            self.snapshot_done.record(self.actor_stream)

    def _collect(self):
        k = (self.rollout._cur + 1) % 2
        with torch.cuda.stream(self.actor_stream):
            self.actor_stream.wait_event(self.batch_free[k])
            batch = self.rollout.collect(self.actor_model)
            self.batch_ready[k].record(self.actor_stream)
        return batch, k

    def prime(self):
        """Collect the first batch so that every later step() has one to learn from."""
        if self.pending is None:
            self._snapshot()
            self.pending = self._collect()

    def step(self, learning_rate, entropy_coeff):
        """Enqueue one learner update on the previously collected batch and, concurrently, the
        collection of the next one.  Returns (vtrace_loss, kl) of the update (device tensors)."""
        self.prime()
        batch, k = self.pending
        # the snapshot for the next rollout is taken first; the learner may not touch the
        # parameters before it is done
        self._snapshot()
        with torch.cuda.stream(self.learn_stream):
            self.learn_stream.wait_event(self.snapshot_done)
            self.learn_stream.wait_event(self.batch_ready[k])
            out = self.alg.learn(batch['obs'], batch['actions'], batch['behaviour_logits'], batch['rewards'],
                                 batch['dones'], learning_rate, entropy_coeff, time_major=True)
            self.weights_ready.record(self.learn_stream)
            self.batch_free[k].record(self.learn_stream)
        for v in batch.values():  # tensors made on the actor stream (e.g. dones.bool()), read on the learner's
            v.record_stream(self.learn_stream)
        self.pending = self._collect()
        return out

    def synchronize(self):
        self.actor_stream.synchronize()
        self.learn_stream.synchronize()
