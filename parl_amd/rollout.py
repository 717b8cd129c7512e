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
import os

import torch

from . import dist as pdist
from . import ops


def _policy_and_draw(model, obs, logits_out, actions_out, seed, offset, row0, offset_base=None):
    """behaviour logits of `obs` into the [E, A] slab and the sampled actions into the [E] slab; the Philox offset
    is `offset` (+ *offset_base on the device, for a model whose policy_sample_into takes one)"""
    if hasattr(model, 'policy_sample_into'):  # head + draw in one launch
        if offset_base is not None:
            model.policy_sample_into(obs, logits_out, actions_out, seed, offset, row0, offset_base=offset_base)
        else:
            model.policy_sample_into(obs, logits_out, actions_out, seed, offset, row0)
        return
    assert offset_base is None
    if hasattr(model, 'policy_into'):
        model.policy_into(obs, logits_out)  # the head's GEMM writes the slab directly
    else:
        logits_out.copy_(model.policy(obs))
    ops.policy_sample_into(logits_out, actions_out, seed, offset, row0)


class RingBatch(object):
    """The observations of a collected rollout, NOT materialised: they are the single frames of ring `index` of the
    env (the reference's actor ships `obs` as a [T*E, 4, d, d] array, examples/IMPALA/actor.py:78-89; four stacked
    copies of every frame).  A learner takes the rows it needs when it needs them: `gather_sequences` (the T rows of
    sequences [b0, b0 + nb), time-major — one 1000-row update of the reference's train_batch_size) or `materialize`
    (the whole [T*E, 4, d, d] batch).  Valid until the env writes that ring again (the rollout after next)."""

    def __init__(self, env, index, T):
        self.env, self.index, self.T = env, int(index), int(T)
        E = env.envs_num
        self.shape = (self.T * E, 4, env.dim, env.dim)
        self.dtype, self.device, self.is_cuda = torch.uint8, env.device, env.device.type == 'cuda'
        self._idx = {}

    def _indices(self, b0, nb):
        key = (int(b0), int(nb))
        hit = self._idx.get(key)
        if hit is None:
            T, dev = self.T, self.device
            slots = (torch.arange(T, dtype=torch.int32, device=dev) + 3).repeat_interleave(nb)
            envs = (torch.arange(nb, dtype=torch.int32, device=dev) + b0).repeat(T)
            hit = self._idx[key] = (slots, envs)
        return hit

    def gather_sequences(self, b0, nb, out=None):
        slots, envs = self._indices(b0, nb)
        return self.env.gather(slots, envs, out, ring_index=self.index)

    def materialize(self, out=None):
        return self.gather_sequences(0, self.env.envs_num, out)

    def record_stream(self, stream):   # the ring is a long-lived tensor of the env, not an allocation of this batch
        pass

    # tensor-like conveniences for host code that inspects a batch (tests, tools): each materialises a fresh tensor
    def clone(self):
        return self.materialize()

    def reshape(self, *shape):
        return self.materialize().reshape(*shape)

    view = reshape

    def cpu(self):
        return self.materialize().cpu()


class DeviceRollout(object):
    def __init__(self, env, sample_batch_steps, seed=0, n_buffers=1, lazy_obs=False):
        """n_buffers > 1: successive collect() calls fill the trajectory slabs round-robin, so a
        learner on another stream can still read batch i-1 while batch i is being written
        (AsyncActorLearner below).  lazy_obs: the batch's 'obs' is a RingBatch (the env gets one frame ring per
        buffer) instead of a materialised [T*E, 4, d, d] tensor per buffer."""
        assert env.horizon >= sample_batch_steps, 'env ring too short for the rollout'
        self.env, self.T, self.seed = env, int(sample_batch_steps), int(seed)
        E, A, dev = env.envs_num, env.act_dim, env.device
        T = self.T
        self.lazy_obs = bool(lazy_obs)
        if self.lazy_obs:
            env.ensure_rings(max(2, int(n_buffers)))
        self._bufs = []
        for _ in range(int(n_buffers)):
            self._bufs.append({
                'actions': torch.zeros((T, E), dtype=torch.int64, device=dev),
                'behaviour_logits': torch.zeros((T, E, A), dtype=torch.float32, device=dev),
                'rewards': torch.zeros((T, E), dtype=torch.float32, device=dev),
                'dones': torch.zeros((T, E), dtype=torch.uint8, device=dev),
            })
            if not self.lazy_obs:
                self._bufs[-1]['obs'] = torch.zeros((T * E, 4, env.dim, env.dim), dtype=torch.uint8, device=dev)
        self._ring_of_buf = [0] * int(n_buffers)   # lazy_obs: the env ring a buffer's rollout was written to
        self._ring_batches = {}
        self._cur = -1
        self._select(0)
        self._obs_step = torch.zeros((E, 4, env.dim, env.dim), dtype=torch.uint8, device=dev)
        self._slots = (torch.arange(T, dtype=torch.int32, device=dev) + 3).repeat_interleave(E)
        self._envs = torch.arange(E, dtype=torch.int32, device=dev).repeat(T)
        self.step_count = 0  # Philox offset: one uniform per (global step, env)
        # the number of the current rollout's first step ON THE DEVICE (collect_begin): a rollout segment replayed as
        # a hipGraph has frozen kernel arguments, its steps draw at offset *base + (step inside the rollout)
        self._step_base = torch.zeros(1, dtype=torch.int64, device=dev)
        self._base_count = 0
        self._graphs, self._graph_pool, self._segment_runs = {}, None, {}
        self._graph_models, self._live_cuts = {}, set()   # models the graphs refer to; the [t0, t1) cuts in use
        # MonitorEnv statistics (atari_wrappers.py:44-100), reduced on the device:
        # (episodes closed, sum of unclipped returns, sum of lengths)
        self.ep_stats = torch.zeros(3, dtype=torch.float64, device=dev)
        self.started = False

    def _select(self, k):
        b = self._bufs[k]
        self.actions, self.behaviour_logits = b['actions'], b['behaviour_logits']
        self.rewards, self.dones, self.obs = b['rewards'], b['dones'], b.get('obs')

    @torch.no_grad()
    def collect_begin(self):
        env = self.env
        self._cur = (self._cur + 1) % len(self._bufs)
        self._select(self._cur)
        if not self.started:
            env.reset()
            self.started = True
        else:
            env.roll()
        self._base_count = self.step_count
        if self._step_base.is_cuda:
            self._step_base.fill_(self.step_count)
        self._ring_of_buf[self._cur] = env.ring_index

    @torch.no_grad()
    def collect_step(self, model, t):
        env = self.env
        # a model whose trunk reads the ring in place gets a reference (one launch less per env step), any other the stack
        obs = env.current_obs_ref(self._obs_step) if getattr(model, 'reads_ring', False) else env.current_obs(self._obs_step)
        logits = self.behaviour_logits[t]
        if self._head_in_env_step(model):
            # policy head + draw at the head of the env launch, the observation at its tail: the step is conv12 ->
            # trunk GEMM -> ONE env launch (DeviceVectorEnv.step_policy_async); same uniforms as below
            h = model.policy_hidden(obs)
            based = self._step_base.is_cuda
            env.step_policy_async(h, model.policy_fc.weight, model.policy_fc.bias, logits, self.actions[t], self.seed,
                                  self.step_count - self._base_count if based else self.step_count, env.env_id0,
                                  self._step_base if based else None, self.rewards[t], self.dones[t],
                                  ep_acc=self.ep_stats)
            self.step_count += 1
            return
        if getattr(model, 'supports_offset_base', False) and self._step_base.is_cuda:
            # the same uniforms as offset = step_count: step_count == *base + (steps since collect_begin)
            _policy_and_draw(model, obs, logits, self.actions[t], self.seed, self.step_count - self._base_count,
                             env.env_id0, self._step_base)
        else:
            _policy_and_draw(model, obs, logits, self.actions[t], self.seed, self.step_count, env.env_id0)
        env.step_async(self.actions[t], self.rewards[t], self.dones[t], ep_acc=self.ep_stats)
        self.step_count += 1

    def _head_in_env_step(self, model):
        """PARL_AMD_FUSED_HEAD=0: head + draw as their own launch (ops.policy_head_sample_into) — A/B and tests"""
        fc = getattr(model, 'policy_fc', None)
        return bool(hasattr(model, 'policy_hidden') and fc is not None and fc.bias is not None and
                    fc.weight.dtype == torch.float32 and hasattr(self.env, 'can_step_policy') and
                    self.env.can_step_policy(fc.in_features, fc.out_features) and
                    os.environ.get('PARL_AMD_FUSED_HEAD', '1') != '0')

    def can_graph(self, model):
        return bool(getattr(model, 'supports_offset_base', False) and self._step_base.is_cuda and self.env.link is None
                    and getattr(model, 'reads_ring', False))

    @torch.no_grad()
    def collect_segment(self, model, t0, t1, graph=False):
        """env steps [t0, t1) of the current rollout.  graph=True (and a model / env that allow it): the segment is
        ONE hipGraph launch — every kernel argument of these steps is a fixed address (ring slots, trajectory slab
        rows of the selected buffer, the actor model's parameters) except the Philox offset, which is read from
        `_step_base`.  The first run of a (buffer, segment) is eager (libraries warm), the second is captured, every
        later one a replay.  A replay enqueues exactly the kernels the eager steps enqueue."""
        if not (graph and self.can_graph(model)):
            for t in range(t0, t1):
                self.collect_step(model, t)
            return
        env = self.env
        # the slabs of buffer _cur, the frames of ring ring_index.  The entry holds a reference to the model (its id
        # cannot be reused by another module while a graph with its parameter addresses is alive), and graphs of
        # segment cuts nobody asks for any more (recalibrated / checkpoint-loaded refresh points) are dropped
        key = (self._cur, env.ring_index, int(t0), int(t1), id(model))
        self._graph_models[id(model)] = model
        if (int(t0), int(t1)) not in self._live_cuts:
            self._live_cuts = {c for c in self._live_cuts if not (c[0] < t1 and t0 < c[1])} | {(int(t0), int(t1))}
            self._drop_graphs(lambda k: (k[2], k[3]) not in self._live_cuts)
        g = self._graphs.get(key)
        if g is None:
            runs = self._segment_runs.get(key, 0)
            self._segment_runs[key] = runs + 1
            if runs == 0:
                for t in range(t0, t1):
                    self.collect_step(model, t)
                return
            assert env.t == t0, (env.t, t0)
            sc, et = self.step_count, env.t
            st = torch.cuda.current_stream(env.device)
            g = torch.cuda.CUDAGraph()
            kw = {'pool': self._graph_pool} if self._graph_pool is not None else {}
            kw.update(pdist.graph_capture_kwargs())
            with torch.cuda.graph(g, stream=st, **kw):
                for t in range(t0, t1):
                    self.collect_step(model, t)
            if self._graph_pool is None:
                self._graph_pool = g.pool()
            self._graphs[key] = g
            self.step_count, env.t = sc, et   # the capture enqueued nothing
        g.replay()
        self.step_count += t1 - t0
        env.t += t1 - t0

    def _drop_graphs(self, stale):
        for k in [k for k in self._graphs if stale(k)]:
            del self._graphs[k]
        for k in [k for k in self._segment_runs if stale(k)]:
            del self._segment_runs[k]

    def _batch_view(self):
        """the time-major batch over the selected trajectory buffer"""
        E = self.env.envs_num
        obs = self.obs
        if self.lazy_obs:
            key = (self._cur, self._ring_of_buf[self._cur])
            obs = self._ring_batches.get(key)
            if obs is None:
                obs = self._ring_batches[key] = RingBatch(self.env, self._ring_of_buf[self._cur], self.T)
        return {
            'obs': obs,
            'actions': self.actions.reshape(self.T * E),
            'behaviour_logits': self.behaviour_logits.reshape(self.T * E, -1),
            'rewards': self.rewards.reshape(self.T * E),
            'dones': self.dones.reshape(self.T * E).bool(),
        }

    @torch.no_grad()
    def collect_end(self):
        if not self.lazy_obs:
            self.env.gather(self._slots, self._envs, self.obs)
        return self._batch_view()

    def collect_steps(self, model):
        for t in range(self.T):
            self.collect_step(model, t)

    def collect(self, model):
        """Run T env steps with `model` as behaviour policy; returns the time-major batch."""
        self.collect_begin()
        self.collect_steps(model)
        return self.collect_end()

    def state_dict(self):
        """the sampler's own state: the Philox offset (one uniform per global step and env), the
        buffer cursor, the episode statistics not yet popped.  The env is saved separately.  The device is
        synchronised first: the statistics may be accumulated on another stream than the caller's."""
        if self.ep_stats.is_cuda:
            torch.cuda.synchronize(self.ep_stats.device)
        return {'step_count': self.step_count, 'started': self.started, 'cur': self._cur, 'seed': self.seed,
                'ep_stats': self.ep_stats.detach().cpu().clone(), 'ring_of_buf': list(self._ring_of_buf),
                'lazy_obs': self.lazy_obs}

    def buffer_state(self, k):
        """the trajectory slabs of buffer k (a collected batch somebody still has to learn from)"""
        return {n: v.detach().cpu().clone() for n, v in self._bufs[k].items()}

    def load_buffer_state(self, k, d):
        missing = [n for n in self._bufs[k] if n not in d]
        if missing:   # e.g. a checkpoint whose observations stayed in the frame rings (lazy_obs) loaded without it
            raise ValueError('trajectory buffer %d: the checkpoint holds no %s (saved with lazy_obs=%s, this rollout has '
                             'lazy_obs=%s: PARL_AMD_LAZY_OBS must be what it was when the checkpoint was written)' %
                             (k, ', '.join(repr(n) for n in missing), 'obs' not in d, self.lazy_obs))
        for n, v in self._bufs[k].items():
            v.copy_(d[n].to(v.device))

    def load_state_dict(self, d):
        if d['seed'] != self.seed:
            raise ValueError('rollout seed %r differs from the checkpoint (%r)' % (self.seed, d['seed']))
        if bool(d.get('lazy_obs', self.lazy_obs)) != self.lazy_obs:
            raise ValueError('the checkpoint was written with lazy_obs=%s (observations %s), this rollout has lazy_obs=%s: '
                             'build the pipeline with the same PARL_AMD_LAZY_OBS' %
                             (d['lazy_obs'], 'in the env\'s frame rings' if d['lazy_obs'] else 'materialised per buffer',
                              self.lazy_obs))
        if self.lazy_obs and ('ring_of_buf' not in d or len(d['ring_of_buf']) != len(self._ring_of_buf)):
            raise ValueError('the checkpoint names the frame ring of %s trajectory buffers, this rollout has %d: a pending '
                             'batch would read whichever ring is current' %
                             (len(d['ring_of_buf']) if 'ring_of_buf' in d else 'no', len(self._ring_of_buf)))
        self.step_count, self.started, self._cur = int(d['step_count']), bool(d['started']), int(d['cur'])
        if 'ring_of_buf' in d and len(d['ring_of_buf']) == len(self._ring_of_buf):
            self._ring_of_buf = [int(x) for x in d['ring_of_buf']]
        if self._cur >= 0:
            self._select(self._cur)
        self.ep_stats.copy_(d['ep_stats'].to(self.ep_stats.device))

    def pop_episode_stats(self):
        """(episodes closed, mean unclipped return, mean length in emulated frames); syncs."""
        n, r, l = (float(x) for x in self.ep_stats.tolist())
        self.ep_stats.zero_()
        return n, (r / n if n else None), (l / n if n else None)


class ElasticDeviceRollout(DeviceRollout):
    """DeviceRollout whose launches are ELASTIC (DeviceVectorEnv.step_elastic_async): a launch never
    emulates more than 4 frames per env.  With synchronous launches the whole vector waits for its
    slowest env, and with 1024 Breakout envs some env is inside the 12-frame life-loss reset
    (EpisodicLifeEnv + FireResetEnv, atari_wrappers.py:200-211, :163-171) in nearly every launch — every
    launch then takes 16 frames instead of 4.  The reference never has this problem: its actors are
    independent processes (examples/IMPALA/train.py:155-194).  Here an env in such a sequence drops out
    of the next launches (no action consumed, no row produced) and the others go on — also across the
    end of a batch: an env that has its T rows starts on the next batch's rows (up to one batch ahead
    of the slowest env), so nobody idles at a batch boundary.  A batch closes when every env has
    STARTED its T rows (a row's reward / done are known in the launch that starts it).
    Every env's rows are exactly the ones the synchronous rollout would have produced for the same
    actions; what changes is only in WHICH launch a row was produced, so per-launch outputs (policy
    logits, sampled actions, frames) live in rings indexed by launch and are compacted through
    row_launch when the batch closes.  env.horizon + 4 = the ring's slots: it must hold the launches of
    two batches (the rows of batch k are produced while batches k-1 and k are open)."""

    def __init__(self, env, sample_batch_steps, seed=0, n_buffers=1, poll_lag=2):
        super(ElasticDeviceRollout, self).__init__(env, sample_batch_steps, seed=seed, n_buffers=n_buffers)
        E, A, dev, T = env.envs_num, env.act_dim, env.device, self.T
        self.S = env.slots  # ring slots == launches the rings remember
        assert self.S >= 3 * T + 16, 'elastic rollout: env horizon + 4 must cover the launches of three batches (see collect_steps)'
        i32 = dict(dtype=torch.int32, device=dev)
        self.actions_lm = torch.zeros((self.S, E), dtype=torch.int64, device=dev)
        self.logits_lm = torch.zeros((self.S, E, A), dtype=torch.float32, device=dev)
        self.rows_done = torch.zeros(E, **i32)
        self.row_launch = torch.zeros((2 * T, E), **i32)
        self.row_slot = torch.zeros((2 * T, E), **i32)
        self.rewards_rows = torch.zeros((2 * T, E), dtype=torch.float32, device=dev)
        self.dones_rows = torch.zeros((2 * T, E), dtype=torch.uint8, device=dev)
        self.finished = torch.zeros(2, **i32)
        self._fin_host = torch.zeros(self.S, dtype=torch.int32)
        if dev.type == 'cuda':
            self._fin_host = self._fin_host.pin_memory()
        self._fin_events = [None] * self.S
        self.poll_lag = int(poll_lag)
        self.launch = 0    # launches since the reset of the run
        self.batch = 0     # batches closed
        self.launches = 0  # launches enqueued by the last collect_steps()
        self._era = [0, 0, 0]  # launch at which the collection of the last three batches began

    @torch.no_grad()
    def collect_begin(self):
        self._cur = (self._cur + 1) % len(self._bufs)
        self._select(self._cur)
        if not self.started:
            self.env.reset()
            self.env.elastic_begin()
            self.started = True

    @torch.no_grad()
    def collect_launch(self, model):
        env, l = self.env, self.launch
        k = l % self.S
        obs = env.elastic_obs(self._obs_step)
        logits = self.logits_lm[k]
        _policy_and_draw(model, obs, logits, self.actions_lm[k], self.seed, self.step_count, env.env_id0)
        # envs may start rows of batch `batch + 1` while batch `batch` is still open, not more
        env.step_elastic_async(self.actions_lm[k], l, (self.batch + 2) * self.T, 2 * self.T, self.T, self.rows_done,
                               self.row_launch, self.row_slot, self.finished, self.rewards_rows, self.dones_rows)
        env.accumulate_episode_stats(self.ep_stats)
        self.step_count += 1
        self.launch += 1

    @torch.no_grad()
    def collect_steps(self, model):
        """launches until every env has started the T rows of the open batch.  The host polls the device's
        completion counter `poll_lag` launches behind its enqueue position, so the GPU never waits for
        the host; the launches enqueued past the closing one already work on the next batch."""
        E, par = self.env.envs_num, self.batch & 1
        st = torch.cuda.current_stream(self.env.device)
        first = self.launch
        self._era = [self._era[1], self._era[2], first]
        polled = first  # launches < polled have been looked at
        while True:
            # What the rings must still hold: the rows of this batch were STARTED since the previous batch opened
            # (an env runs at most one batch ahead), but the observation a row acts on — and its three linked
            # FrameStack predecessors — can be older: an env that finished this batch's rows early idled at
            # rows_limit with an observation produced while the batch BEFORE the previous one was collected.
            # So the oldest slot still referenced is younger than era[0] - 3 - (launches an env sat out in a
            # reset sequence, <= 4): three eras + 8 slots of margin.
            if self.launch - self._era[0] >= self.S - 8:
                raise RuntimeError('elastic rollout: %d launches since the batch before the previous one opened '
                                   'exceed the ring (%d slots): raise the env horizon' %
                                   (self.launch - self._era[0], self.S))
            self.collect_launch(model)
            l = self.launch - 1
            self._fin_host[l % self.S:l % self.S + 1].copy_(self.finished[par:par + 1], non_blocking=True)
            ev = self._fin_events[l % self.S] or torch.cuda.Event()
            self._fin_events[l % self.S] = ev
            ev.record(st)
            done = False
            while polled <= l - self.poll_lag:
                self._fin_events[polled % self.S].synchronize()
                if int(self._fin_host[polled % self.S]) >= E:
                    done = True
                    break
                polled += 1
            if done:
                break
        self.launches = self.launch - first

    @torch.no_grad()
    def collect_end(self):
        env, T = self.env, self.T
        E = env.envs_num
        h = (self.batch & 1) * T
        rl = self.row_launch[h:h + T]
        idx = (rl % self.S).long()  # [T, E]: ring index of the launch that started env e's row r
        torch.gather(self.actions_lm, 0, idx, out=self.actions)
        torch.gather(self.logits_lm, 0, idx[:, :, None].expand(T, E, self.logits_lm.shape[2]), out=self.behaviour_logits)
        env.gather(self.row_slot[h:h + T].reshape(-1), self._envs, self.obs)
        self.rewards.copy_(self.rewards_rows[h:h + T])
        self.dones.copy_(self.dones_rows[h:h + T])
        self.finished[self.batch & 1].zero_()  # counts batch + 2 next; no env reaches its last row before the limit moves
        self.batch += 1
        return self._batch_view()

    def collect(self, model):
        self.collect_begin()
        self.collect_steps(model)
        return self.collect_end()

    def collect_step(self, model, t):
        raise RuntimeError('ElasticDeviceRollout has no fixed step count: use collect_steps()')

    def state_dict(self):
        d = super(ElasticDeviceRollout, self).state_dict()
        d.update({'launch': self.launch, 'batch': self.batch, 'era': list(self._era),
                  'elastic': {k: getattr(self, k).detach().cpu().clone()
                              for k in ('actions_lm', 'logits_lm', 'rows_done', 'row_launch', 'row_slot',
                                        'rewards_rows', 'dones_rows', 'finished')}})
        return d

    def load_state_dict(self, d):
        super(ElasticDeviceRollout, self).load_state_dict(d)
        self.launch, self.batch, self._era = int(d['launch']), int(d['batch']), ([0] + list(d['era']))[-3:]
        for k, v in d['elastic'].items():
            getattr(self, k).copy_(v.to(getattr(self, k).device))


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
        self.ep_stats = torch.zeros(3, dtype=torch.float64, device=dev)
        self.started = False
        self.adv = torch.zeros((T, E), dtype=torch.float32, device=dev)
        self.target = torch.zeros((T, E), dtype=torch.float32, device=dev)
        self._step_base = torch.zeros(1, dtype=torch.int64, device=dev)   # number of the rollout's first step (hipGraph replays)
        self._graphs, self._runs, self._side, self._graph_models = {}, {}, None, {}

    def _steps(self, model, base=None):
        """the T steps of a rollout, the bootstrap value forward, the GAE launch and the batch's observations; with
        `base` (int64 [1] on the device = the number of the rollout's first step) the Philox offsets are
        base + t — what a hipGraph of this body needs"""
        env = self.env
        ring = getattr(model, 'reads_ring', False)   # a trunk that reads the ring in place: one launch less per step
        for t in range(self.T):
            obs = env.current_obs_ref(self._obs_step) if ring else env.current_obs(self._obs_step)
            logits, values = model.policy_and_value(obs)
            self.values[t].copy_(values)
            if base is not None:
                ops.policy_sample_into(logits, self.actions[t], self.seed, t, env.env_id0, offset_base=base)
            else:
                ops.policy_sample_into(logits, self.actions[t], self.seed, self.step_count + t, env.env_id0)
            env.step_async(self.actions[t], self.rewards[t], self.dones[t], ep_acc=self.ep_stats)
        next_value = model.value(env.current_obs_ref(self._obs_step) if ring else env.current_obs(self._obs_step))  # ignored where the last step was terminal
        adv, target = ops.gae(self.rewards, self.values, self.dones, next_value, self.gamma, self.lam)
        self.adv.copy_(adv)
        self.target.copy_(target)
        env.gather(self._slots, self._envs, self.obs)

    def _can_graph(self, model):
        """the rollout as ONE hipGraph — opt-in (PARL_AMD_A2C_GRAPH=1): measured neutral on configs[1] (256 envs, T = 20:
        984 k frames/s graphed, 998 k eager on one box — the host already runs ahead of a 0.8 ms env step), so eager
        launches stay the default.  Needs a model that keeps its operand-order weight copies at fixed addresses and
        refreshes them on request (AtariModel84) and a device env; identical batches (test_gpu_examples.py)"""
        return bool(self.env.device.type == 'cuda' and hasattr(model, 'refresh_actor_layout') and
                    hasattr(model, '_lay_pinned') and os.environ.get('PARL_AMD_A2C_GRAPH', '0') == '1')

    @torch.no_grad()
    def collect(self, model):
        env = self.env
        if not self.started:
            env.reset()
            self.started = True
        else:
            env.roll()
        n = self.T * env.envs_num
        batch = {'obs': self.obs, 'actions': self.actions.reshape(n), 'advantages': self.adv.reshape(n),
                 'target_values': self.target.reshape(n)}
        key = id(model)
        self._graph_models[key] = model   # the graph holds its parameter addresses: the id must not be reused while it lives
        g = self._graphs.get(key) if self._can_graph(model) else None
        runs = self._runs.get(key, 0)
        self._runs[key] = runs + 1
        if g is None and not (self._can_graph(model) and runs >= 1):
            self._steps(model)           # the first rollout (libraries warm), or a model / env that cannot be graphed
            self.step_count += self.T
            return batch
        cur = torch.cuda.current_stream(env.device)
        if g is None:
            # the second rollout is captured: every kernel argument of a rollout is a fixed address (ring slots, the
            # [T,E] slabs, the model's parameters and its operand-order weight buffers) except the Philox offset
            side = self._side = torch.cuda.Stream(device=env.device)
            t0 = env.t
            g = torch.cuda.CUDAGraph()
            side.wait_stream(cur)
            with torch.cuda.graph(g, stream=side, **pdist.graph_capture_kwargs()):
                model.refresh_actor_layout()     # inside the graph: every replay starts from the current weights
                model._lay_pinned = True
                try:
                    self._steps(model, base=self._step_base)
                finally:
                    model._lay_pinned = False
            cur.wait_stream(side)
            env.t = t0                           # the capture enqueued nothing
            self._graphs[key] = g
        self._step_base.fill_(self.step_count)
        g.replay()
        env.t += self.T
        self.step_count += self.T
        return batch

    pop_episode_stats = DeviceRollout.pop_episode_stats

    def state_dict(self):
        return {'step_count': self.step_count, 'started': self.started, 'seed': self.seed,
                'ep_stats': self.ep_stats.detach().cpu().clone()}

    def load_state_dict(self, d):
        if d['seed'] != self.seed:
            raise ValueError('rollout seed %r differs from the checkpoint (%r)' % (self.seed, d['seed']))
        self.step_count, self.started = int(d['step_count']), bool(d['started'])
        self.ep_stats.copy_(d['ep_stats'].to(self.ep_stats.device))


class _GraphedLoss(object):
    """the VTraceLoss attributes of the last graphed update (views of GraphedLearn.out, float64 on the device)"""

    def __init__(self, out):
        self.total_loss, self.pi_loss, self.vf_loss, self.entropy = out[0], out[1], out[2], out[3]
        self.vtrace_returns = None


def fixed_refresh_points(T, updates_per_rollout, obs_dim):
    """The default mid-rollout weight-refresh points of AsyncActorLearner: at env steps T/5, 2T/5, 3T/5 the
    actors pick up what the learner published after a fixed share of the concurrent pass's updates.  The share
    is what the learner has safely passed by then on an MI355X — at 42x42 its 51 updates take the time of ~20
    env steps (28 % of the updates per fifth of the rollout is ~60 % of what it has done), at 84x84 learner
    pass and rollout take equally long (9 % per fifth) — and depends on nothing measured at run time, so the
    same command line is the same training run everywhere.  A slower learner makes the actors wait at a
    point, never read a half-written publication (events order the two streams)."""
    n, T = int(updates_per_rollout), int(T)
    share = 0.28 if obs_dim <= 42 else 0.09
    pts, last = [], 0
    for k in (1, 2, 3):
        s_, u = (k * T) // 5, min(n, int(k * share * n))
        if 0 < s_ < T and u > last and u >= 2:
            pts.append((s_, u))
            last = u
    return pts


class AsyncActorLearner(object):
    """IMPALA's actor / learner decoupling (examples/IMPALA/train.py:155-194: sample threads fill a
    queue while the learn thread drains it; actors act with parameters that lag the learner by up
    to `params_broadcast_interval` updates and V-trace corrects for the lag) on ONE GPU:
    HIP streams instead of threads and processes.

      actor stream g  weights snapshot -> T env steps of env group g (policy fwd, sample, emulator,
                      frame_post) -> batch i of group g
      learner stream  ONE update on the union of the groups' batches i-1 (fwd, fused V-trace, bwd
                      per group, accumulated; [all-reduce], clip, Adam once)

    All of it is enqueued by one host thread; events order the streams.  The emulator kernel keeps
    one wavefront pair per env busy and is latency-bound, so (a) the learner's GEMMs run in the issue
    slots it leaves free, and (b) with G >= 2 env groups the policy forward / sampling of one
    group runs while the other groups' emulator kernels are in flight (the reference gets the
    same effect from its 32 independent actor processes).  The actors' parameter snapshot plays
    the role of the reference actor's `set_weights` (actor.py:103-104): it is refreshed from the
    learner before every rollout, so the behaviour policy lags the learner by exactly one update."""

    def __init__(self, alg, envs, sample_batch_steps, seed=0, elastic=False, train_batch_size=None,
                 refresh_points='fixed'):
        """elastic: ElasticDeviceRollout (one env group; the env's horizon is the launch bound of a batch).
        train_batch_size: the reference's learner batch in ROWS (impala_config.py:31: 1000 = 20 sequences of
        T = 50; the learner concatenates actor batches until it holds at least that many, train.py:98).  None:
        one update per step() on the whole T*E rollout.  Otherwise step() runs E // (train_batch_size // T)
        updates, each on the next `train_batch_size // T` sequences of the rollout (the last one takes the
        remainder as well), every update one hipGraph replay (algorithms.impala.graphed.GraphedLearn).
        refresh_points (train_batch_size mode): the actors' mid-rollout weight refresh, see below — 'fixed'
        (the default: `fixed_refresh_points`, a function of T, the updates per rollout and the frame size only,
        so a run is the same run on every box), 'auto' (calibrated once from the measured speed of learner pass
        and rollout: adapts to the machine, not reproducible across machines), an explicit list of (env step,
        updates of the concurrent pass done) pairs, or None / [] (off)."""
        import copy
        self.alg = alg
        self.envs = list(envs) if isinstance(envs, (list, tuple)) else [envs]
        self.env = self.envs[0]
        self.T = int(sample_batch_steps)
        # one Philox key for all groups: streams are told apart by the global env id
        if elastic and len(self.envs) != 1:
            raise ValueError('elastic rollouts take one env group')
        cls = ElasticDeviceRollout if elastic else DeviceRollout
        lazy = not elastic and bool(int(os.environ.get('PARL_AMD_LAZY_OBS', '1')))   # obs stay in the frame rings (RingBatch)
        self.rollouts = [cls(e, sample_batch_steps, seed=seed, n_buffers=2, **({'lazy_obs': True} if lazy else {})) for e in self.envs]
        self._obs_full = {}   # one materialised [T*E, 4, d, d] batch per env group for the one-update-per-rollout mode
        self.rollout = self.rollouts[0]
        self.actor_model = copy.deepcopy(alg.model)
        for p in self.actor_model.parameters():
            p.requires_grad_(False)
        dev = self.env.device
        # the rollouts are the critical path (latency-bound emulator): high-priority queues
        ap, lp = (int(x) for x in os.environ.get('PARL_AMD_STREAM_PRIO', '-1,0').split(','))
        self.actor_streams = [torch.cuda.Stream(device=dev, priority=ap) for _ in self.envs]
        self.actor_stream = self.actor_streams[0]
        self.learn_stream = torch.cuda.Stream(device=dev, priority=lp)
        self.weights_ready = torch.cuda.Event()
        self.snapshot_done = torch.cuda.Event()
        G = len(self.envs)
        self.batch_ready = [[torch.cuda.Event(), torch.cuda.Event()] for _ in range(G)]
        self.batch_free = [torch.cuda.Event(), torch.cuda.Event()]
        self.pending = None  # ([batch per group], buffer index) collected but not yet learned
        # data-parallel runs: all-gather the small per-step tensors of every learned batch on the
        # LEARNER stream, right after the gradient all-reduce (one stream, one fixed order of
        # collectives on every rank)
        self.gather_small = False
        self.gathered = None
        self.step_done = torch.cuda.Event()  # recorded after every update on the learner stream
        self.sub_batches = None
        self.updates = 0  # parameter updates enqueued so far
        if train_batch_size:
            if len(self.envs) != 1:
                raise ValueError('train_batch_size takes one env group')
            E = self.env.envs_num
            seqs = max(1, int(train_batch_size) // self.T)
            n = max(1, E // seqs)
            self.sub_batches = [(i * seqs, seqs if i < n - 1 else E - i * seqs) for i in range(n)]
        # Mid-rollout weight refresh (train_batch_size mode, synchronous launches).  With n updates per rollout
        # the actors' snapshot is n .. 2n updates old by the time its rows are learned from (1024 envs: 51 ..
        # 102; the reference's 32 staggered actors fetch the current weights before every 50-step sample,
        # train.py:176-191: a handful).  At 42x42 the learner is much faster than the rollout (its 51 updates take
        # the time of ~20 env steps), so the actors pick newer weights up INSIDE the rollout: at env step s they
        # wait for (an event after) update u(s) of the concurrent pass and copy what the learner published
        # there.  u(s) must be something the learner has long passed when the actors reach step s — at 84x84
        # learner and rollout take equally long, waiting for "40 % of the updates at 20 % of the rollout"
        # stalled the actors (1.67 -> 1.36 M frames/s) — so 'auto' measures both once (the second step's learner
        # pass and rollout, HIP events) and asks at steps T/5, 2T/5, 3T/5 for 60 % of the updates the learner is
        # then expected to have done.  [(env step, updates of the concurrent learner pass done)]
        self.refresh_points = []
        self._refresh_auto = False
        self._calib = None
        can = bool(self.sub_batches and not elastic and self.T >= 10 and int(os.environ.get('PARL_AMD_REFRESH', '1')))
        if can and refresh_points == 'auto':
            self._refresh_auto = True
        elif can and refresh_points == 'fixed':
            self.refresh_points = fixed_refresh_points(self.T, len(self.sub_batches), self.env.dim)
        elif can and refresh_points:
            self.refresh_points = self._checked_refresh_points(refresh_points)
        self._src = [p for p in alg.model.parameters()] + [b for b in alg.model.buffers()]
        self._dst = [p for p in self.actor_model.parameters()] + [b for b in self.actor_model.buffers()]
        cur = torch.cuda.current_stream(dev)
        for st in self.actor_streams:
            st.wait_stream(cur)  # env construction (reset cache, tables) ran on `cur`
        self.learn_stream.wait_stream(cur)
        self.weights_ready.record(cur)
        for e in self.batch_free:
            e.record(cur)
        n_pub = 3 if self._refresh_auto else len(self.refresh_points)
        self._pub = [[t.detach().clone() for t in self._src] for _ in range(n_pub)]
        self._pub_ready = [torch.cuda.Event() for _ in range(n_pub)]
        self._pass_enqueued = False  # a learner pass (with its publications) was enqueued before this rollout
        # the actors' rollout as hipGraph segments (between the refresh points): one host call per segment instead of
        # five launches per env step
        self.graph_rollout = bool(int(os.environ.get('PARL_AMD_ROLLOUT_GRAPH', '1'))) and not elastic and len(self.envs) == 1
        self.graphed = {}
        if self.sub_batches:
            from .algorithms.impala.graphed import GraphedLearn
            pool = None
            with torch.cuda.stream(self.learn_stream):
                for _, nb in self.sub_batches:
                    if nb not in self.graphed:
                        self.graphed[nb] = GraphedLearn(alg, nb, (4, self.env.dim, self.env.dim), self.env.act_dim,
                                                        pool=pool)
                        pool = self.graphed[nb].pool
            self.learn_stream.synchronize()

    def _checked_refresh_points(self, points):
        """(env step, updates done) pairs a rollout can honour: 0 < step < T, and an update count some pass
        reaches — waiting for a publication that is never recorded is a no-op on the device, the actors would
        copy whatever the publication buffer held"""
        pts = sorted((int(a), int(b)) for a, b in points)
        n = len(self.sub_batches or ())
        for a, b in pts:
            if not (0 < a < self.T and 0 < b <= n):
                raise ValueError('refresh point (%d, %d): need 0 < env step < %d and 0 < updates <= %d' % (a, b, self.T, n))
        if len(set(a for a, _ in pts)) != len(pts):
            raise ValueError('refresh points: one point per env step')
        return pts

    def _calibrate_refresh(self):
        """'auto' refresh points: step 1 runs plain (graphs and caches warm up), step 2 is timed (learner pass and
        rollout, overlapped as always), step 3 reads the two durations (one host wait, once) and fixes the points"""
        self._steps_seen = getattr(self, '_steps_seen', 0) + 1
        if self._steps_seen == 2:
            self._calib = {}
        elif self._steps_seen == 3 and self._calib and 'r1' in self._calib and 'l1' in self._calib:
            self._calib['r1'].synchronize()
            self._calib['l1'].synchronize()
            L = self._calib['l0'].elapsed_time(self._calib['l1'])
            R = self._calib['r0'].elapsed_time(self._calib['r1'])
            n, T = len(self.sub_batches), self.T
            pts, last = [], 0
            for s_ in (T // 5, (2 * T) // 5, (3 * T) // 5):
                u = min(n, int(0.6 * n * (s_ * R / T) / max(L, 1e-6)))
                if u >= max(2, n // 10) and u > last:
                    pts.append((s_, u))
                    last = u
                if last >= n:
                    break
            self.refresh_points = pts
            self.refresh_calibration = {'learner_pass_ms': L, 'rollout_ms': R}
            self._calib = None
            self._refresh_auto = False

    def _learn_sub_batches(self, batch, learning_rate, entropy_coeff):
        """the rollout as E // seqs updates of the reference's train_batch_size (on the learner stream);
        learning_rate / entropy_coeff: floats, or schedulers whose step() is called once per update as the
        reference's learner does (train.py:111-112)"""
        E = self.env.envs_num
        gl = None
        ls = torch.cuda.current_stream(self.env.device)
        if self._refresh_auto and self._calib is not None and 'l0' not in self._calib:
            self._calib['l0'] = torch.cuda.Event(enable_timing=True)
            self._calib['l0'].record(ls)
        for u, (b0, nb) in enumerate(self.sub_batches):
            gl = self.graphed[nb]
            gl.load(batch, b0, E)
            lr = learning_rate.step() if hasattr(learning_rate, 'step') else learning_rate
            ec = entropy_coeff.step() if hasattr(entropy_coeff, 'step') else entropy_coeff
            gl.replay(lr, ec)
            self.updates += 1
            for i, (_, after) in enumerate(self.refresh_points):
                if after == u + 1:  # publish the weights for the actors' mid-rollout refresh
                    with torch.no_grad():
                        torch._foreach_copy_(self._pub[i], self._src)
                    self._pub_ready[i].record(ls)
        if self._refresh_auto and self._calib is not None and 'l1' not in self._calib:
            self._calib['l1'] = torch.cuda.Event(enable_timing=True)
            self._calib['l1'].record(ls)
        self._pass_enqueued = True
        return _GraphedLoss(gl.out), gl.out[4]

    def pop_learn_stats(self):
        """train_batch_size mode: means of (total_loss, pi_loss, vf_loss, entropy, kl) over the updates since the
        last call and their number (one D2H per log interval; the reference's agent.learn returns them per
        update, atari_agent.py:40-41).  Waits for the learner stream."""
        self.learn_stream.synchronize()
        tot, n = [0.0] * 5, 0
        for gl in self.graphed.values():
            m, k = gl.pop_stats()
            if k:
                tot = [a + b * k for a, b in zip(tot, m)]
                n += k
        return ([x / n for x in tot] if n else None), n

    def _snapshot(self):
        """actor parameters <- learner parameters (the reference actor's set_weights)"""
        s0 = self.actor_streams[0]
        k = self.rollouts[0]._cur
        with torch.cuda.stream(s0):
            s0.wait_event(self.weights_ready)
            if k >= 0:  # every group must be done acting with the old snapshot
                for g in range(len(self.envs)):
                    s0.wait_event(self.batch_ready[g][k])
            with torch.no_grad():
                torch._foreach_copy_(self._dst, self._src)
                self._refresh_actor_layout()
            self.snapshot_done.record(s0)

    def _refresh_actor_layout(self):
        """derived copies of the actors' weights (a model's MFMA operand-order buffers) follow every weight copy, on
        its stream"""
        f = getattr(self.actor_model, 'refresh_actor_layout', None)
        if f is not None:
            f()

    def _collect(self):
        k = (self.rollouts[0]._cur + 1) % 2
        for st, ro in zip(self.actor_streams, self.rollouts):
            with torch.cuda.stream(st):
                st.wait_event(self.snapshot_done)
                st.wait_event(self.batch_free[k])
                ro.collect_begin()
        if self._refresh_auto and self._calib is not None and 'r0' not in self._calib and self._pass_enqueued:
            # calibration rollout: timed, not refreshed
            st = self.actor_streams[0]
            with torch.cuda.stream(st):
                self._calib['r0'] = torch.cuda.Event(enable_timing=True)
                self._calib['r0'].record(st)
                self.rollouts[0].collect_steps(self.actor_model)
                self._calib['r1'] = torch.cuda.Event(enable_timing=True)
                self._calib['r1'].record(st)
            self._pass_enqueued = False
        elif len(self.rollouts) == 1 and self.refresh_points and self._pass_enqueued:
            st, ro = self.actor_streams[0], self.rollouts[0]
            at = {step: i for i, (step, _) in enumerate(self.refresh_points)}
            cuts = sorted(at) + [self.T]
            with torch.cuda.stream(st):
                ro.collect_segment(self.actor_model, 0, cuts[0], graph=self.graph_rollout)
                for a, b in zip(cuts[:-1], cuts[1:]):
                    # the learner pass enqueued before this rollout published these weights
                    st.wait_event(self._pub_ready[at[a]])
                    with torch.no_grad():
                        torch._foreach_copy_(self._dst, self._pub[at[a]])
                        self._refresh_actor_layout()
                    ro.collect_segment(self.actor_model, a, b, graph=self.graph_rollout)
            self._pass_enqueued = False
        elif len(self.rollouts) == 1 and not isinstance(self.rollouts[0], ElasticDeviceRollout):
            with torch.cuda.stream(self.actor_streams[0]):
                self.rollouts[0].collect_segment(self.actor_model, 0, self.T, graph=self.graph_rollout)
        elif len(self.rollouts) == 1:
            with torch.cuda.stream(self.actor_streams[0]):
                self.rollouts[0].collect_steps(self.actor_model)
        else:
            # step-interleaved enqueue: every group's stream always has work queued
            for t in range(self.T):
                for st, ro in zip(self.actor_streams, self.rollouts):
                    with torch.cuda.stream(st):
                        ro.collect_step(self.actor_model, t)
        batches = []
        for g, (st, ro) in enumerate(zip(self.actor_streams, self.rollouts)):
            with torch.cuda.stream(st):
                batches.append(ro.collect_end())
                self.batch_ready[g][k].record(st)
        return batches, k

    def prime(self):
        """Collect the first batch so that every later step() has one to learn from."""
        if self.pending is None:
            self._snapshot()
            self.pending = self._collect()

    def step(self, learning_rate, entropy_coeff):
        """Enqueue one learner update on the previously collected batches and, concurrently, the
        collection of the next ones.  Returns (vtrace_loss, kl) of the update: device tensors
        produced on the LEARNER stream — before reading them from another stream call
        `wait_outputs()` (or `synchronize()`)."""
        self.prime()
        batches, k = self.pending
        if self._refresh_auto:
            self._calibrate_refresh()
        # the snapshot for the next rollout is taken first; the learner may not touch the
        # parameters before it is done
        self._snapshot()
        ls = self.learn_stream
        with torch.cuda.stream(ls):
            ls.wait_event(self.snapshot_done)
            for g in range(len(batches)):
                ls.wait_event(self.batch_ready[g][k])
            if self.sub_batches:
                out = self._learn_sub_batches(batches[0], learning_rate, entropy_coeff)
            elif len(batches) == 1:
                b = self._materialized(batches[0], 0)
                self.updates += 1
                out = self.alg.learn(b['obs'], b['actions'], b['behaviour_logits'], b['rewards'], b['dones'],
                                     learning_rate, entropy_coeff, time_major=True)
            else:
                self.updates += 1
                out = self.alg.learn_batches([self._materialized(b, g) for g, b in enumerate(batches)], learning_rate,
                                             entropy_coeff, time_major=True)
            if self.gather_small:  # per-step scalars of the batch just learned, for global statistics
                from . import dist as pdist
                self.gathered = [pdist.all_gather_small({'rewards': b['rewards'], 'dones': b['dones'].to(torch.uint8),
                                                         'actions': b['actions']}, slot=g)
                                 for g, b in enumerate(batches)]
            self.weights_ready.record(ls)
            self.batch_free[k].record(ls)
            self.step_done.record(ls)
        for b in batches:
            for v in b.values():  # tensors made on an actor stream (e.g. dones.bool()), read on the learner's
                v.record_stream(ls)
        self.pending = self._collect()
        return out

    def _materialized(self, batch, g):
        """a batch whose observations are still in the env's ring (RingBatch) with the stacks gathered — on the
        LEARNER stream, into ONE buffer per env group (the ring keeps the frames until the rollout after next)"""
        if not isinstance(batch['obs'], RingBatch):
            return batch
        buf = self._obs_full.get(g)
        if buf is None:
            buf = self._obs_full[g] = torch.empty(batch['obs'].shape, dtype=torch.uint8, device=batch['obs'].device)
        b = dict(batch)
        b['obs'] = batch['obs'].materialize(buf)
        return b

    def state_dict(self):
        """Everything of the PIPELINE a resumed run needs next to the model / optimizer the caller saves
        (Agent.save): the envs, the samplers, the actors' weight snapshot, and the batch that was collected
        but not learned yet (`pending`: the learner is always one batch behind the actors).  All actor and
        learner streams are drained first — they run launches ahead of the host — so nothing is torn."""
        self.synchronize()
        torch.cuda.synchronize(self.env.device)
        d = {'envs': [e.state_dict() for e in self.envs], 'rollouts': [r.state_dict() for r in self.rollouts],
             'actor_model': {k: v.detach().cpu().clone() for k, v in self.actor_model.state_dict().items()},
             'updates': self.updates, 'pending': None,
             'refresh_points': [list(x) for x in self.refresh_points], 'refresh_auto': bool(self._refresh_auto)}
        if self.pending is not None:
            k = self.pending[1]
            d['pending'] = {'k': k, 'buffers': [r.buffer_state(k) for r in self.rollouts]}
        return d

    def load_state_dict(self, d):
        pts = None
        if 'refresh_points' in d and not d.get('refresh_auto', False):
            # a run resumes with ITS points (checked like the constructor's, before anything is touched): a resume
            # that silently fell back to this object's points would no longer be the same run
            pts = self._checked_refresh_points(d['refresh_points']) if d['refresh_points'] else []
            if len(pts) > len(self._pub):   # publication buffers for the checkpoint's points
                self._pub += [[t.detach().clone() for t in self._src] for _ in range(len(pts) - len(self._pub))]
                self._pub_ready += [torch.cuda.Event() for _ in range(len(pts) - len(self._pub_ready))]
        elif d.get('refresh_auto', False) and not self._refresh_auto:
            raise ValueError('the checkpoint was saved while its refresh points were still being calibrated '
                             "(refresh_points='auto'): resume it into a pipeline built with refresh_points='auto'")
        self.synchronize()
        torch.cuda.synchronize(self.env.device)
        for e, s in zip(self.envs, d['envs']):
            e.load_state_dict(s)
        for r, s in zip(self.rollouts, d['rollouts']):
            r.load_state_dict(s)
        self.actor_model.load_state_dict(d['actor_model'])
        self.updates = int(d['updates'])
        if pts is not None:
            self.refresh_points = pts
            self._refresh_auto, self._calib = False, None
        self._pass_enqueued = False
        self.pending = None
        if d['pending'] is not None:
            k = int(d['pending']['k'])
            for r, b in zip(self.rollouts, d['pending']['buffers']):
                r.load_buffer_state(k, b)
                r._select(k)
            self.pending = ([r._batch_view() for r in self.rollouts], k)
        torch.cuda.synchronize(self.env.device)
        cur = torch.cuda.current_stream(self.env.device)
        for ev in self.batch_free + [self.weights_ready, self.snapshot_done] + [e for pair in self.batch_ready for e in pair]:
            ev.record(cur)

    def pop_episode_stats(self):
        """(episodes closed, mean unclipped return, mean length) over all groups.  The statistics
        are accumulated on the actor streams: the caller's stream first waits for everything
        enqueued there (so the counts are complete and the reset cannot race with an accumulation
        in flight), and the actor streams wait for the reset before they accumulate again."""
        cur = torch.cuda.current_stream(self.env.device)
        for st in self.actor_streams:
            cur.wait_stream(st)
        n = r = l = 0.0
        for ro in self.rollouts:
            gn, gr, gl = ro.pop_episode_stats()
            if gn:
                n, r, l = n + gn, r + gr * gn, l + gl * gn
        for st in self.actor_streams:
            st.wait_stream(cur)
        return n, (r / n if n else None), (l / n if n else None)

    def wait_outputs(self):
        """make the caller's current stream wait for the last update's outputs (loss, kl, gathered)"""
        torch.cuda.current_stream(self.env.device).wait_event(self.step_done)

    def synchronize(self):
        for st in self.actor_streams:
            st.synchronize()
        self.learn_stream.synchronize()
