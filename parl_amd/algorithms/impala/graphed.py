"""IMPALA.learn at the REFERENCE's learner batch as one hipGraph launch.

The reference's learner consumes `train_batch_size` = 1000 rows per update (20 sequences of
sample_batch_steps = 50: examples/IMPALA/impala_config.py:26-31, train.py:90-118).  A 1000-row update is
≈ 0.2 ms of GPU work behind ≈ 100 kernel launches: issued one by one from Python it takes ≈ 2.6 ms and
the learner, not the 1024 on-device actors, bounds the pipeline (51 updates per 51,200-row rollout).
The shapes are static, so the whole update — trunk forward, heads + V-trace loss kernel, backward,
global-norm clip, Adam — is captured ONCE into a HIP graph (torch.cuda.CUDAGraph on ROCm = hipGraph)
and replayed per update: one host call, no Python between the kernels.

  * inputs live in static buffers ([T, B] time-major); `load()` copies a [T, b0:b0+B] slice of a
    rollout's time-major slabs into them (5 strided copies);
  * the learning rate is a DEVICE scalar (Adam `capturable`, tensor lr): a piecewise schedule
    (impala_config.py:34-36) is one `fill_` when the value changes, no re-capture; the entropy
    coefficient is a kernel argument — a new value re-captures (the reference's schedule is constant);
  * with a data-parallel grad_hook (parl_amd.dist.FlatGradAllReduce) over RCCL the all-reduce is captured INSIDE
    the graph (forward + backward into the flat bucket | ncclAllReduce | clip + Adam: still one launch per
    update, round 6); over gloo (ranks sharing a test GPU) or with PARL_AMD_GRAPH_ALLREDUCE=0 the update is TWO
    graphs with the eager collective between them;
  * the four loss terms + KL of every replay are accumulated on the device (`pop_stats()` = means
    since the last pop: one D2H per log interval instead of one per update, atari_agent.py:40-41).

Same arithmetic as IMPALA.learn(time_major=True): the graph replays exactly the kernels that call
issues (tests/test_gpu_graphed.py compares parameters and losses of both over several updates).
"""
import os

import torch

from ... import ops

__all__ = ['GraphedLearn', 'make_capturable', 'load_optimizer_state_inplace']


def make_capturable(optimizer, device):
    """Adam whose step can be captured: step counters on the device, lr a device scalar, the fused
    multi-tensor kernel (one launch for all parameters)."""
    for g in optimizer.param_groups:
        g['capturable'] = True
        g['fused'] = True
        g['foreach'] = False
        if not isinstance(g['lr'], torch.Tensor):
            g['lr'] = torch.tensor(float(g['lr']), dtype=torch.float32, device=device)
        elif g['lr'].device != torch.device(device) or g['lr'].dtype != torch.float32:
            # a checkpoint loaded with map_location='cpu' brings its lr tensor along
            g['lr'] = g['lr'].detach().to(device=device, dtype=torch.float32)
            g.pop('_lr_value', None)
    for st in optimizer.state.values():
        if 'step' in st and (not st['step'].is_cuda or st['step'].dtype != torch.float32):
            st['step'] = st['step'].to(device=device, dtype=torch.float32)


def set_lr(optimizer, lr):
    for g in optimizer.param_groups:
        if isinstance(g['lr'], torch.Tensor):
            if g.get('_lr_value') != lr:
                g['lr'].fill_(lr)
                g['_lr_value'] = lr
        else:
            g['lr'] = lr


def load_optimizer_state_inplace(optimizer, state_dict):
    """optimizer.load_state_dict for an optimizer whose state a captured graph refers to BY ADDRESS: the
    checkpoint's tensors are copied into the existing state tensors (torch's load_state_dict replaces them,
    after which a replay would keep updating the old ones).  The learning rate is restored into the device
    scalar.  State that does not exist yet (no step taken) is created by a normal load."""
    if not optimizer.state:
        _plain_load(optimizer, state_dict)
        dev = next(p for g in optimizer.param_groups for p in g['params']).device
        for g in optimizer.param_groups:  # the checkpoint's memo of the last filled value says nothing about
            g.pop('_lr_value', None)      # the tensor just installed
        make_capturable(optimizer, dev)
        return
    params = [p for g in optimizer.param_groups for p in g['params']]
    ids = [i for g in state_dict['param_groups'] for i in g['params']]
    assert len(ids) == len(params), 'optimizer / checkpoint parameter counts differ'
    with torch.no_grad():
        for p, i in zip(params, ids):
            src, dst = state_dict['state'].get(i, {}), optimizer.state[p]
            for k, v in dst.items():
                if isinstance(v, torch.Tensor):
                    if k in src:
                        v.copy_(torch.as_tensor(src[k]).to(device=v.device, dtype=v.dtype))
                    else:  # saved before this parameter's first step: its state is "no step taken"
                        v.zero_()
        for g, sg in zip(optimizer.param_groups, state_dict['param_groups']):
            lr = sg['lr']
            set_lr(optimizer if len(optimizer.param_groups) == 1 else _one_group(g), float(lr))


class _one_group(object):
    def __init__(self, g):
        self.param_groups = [g]


def _plain_load(optimizer, state_dict):
    load = getattr(optimizer, '_parl_plain_load_state_dict', None) or optimizer.load_state_dict
    load(state_dict)


def _guard_load_state_dict(optimizer):
    """After a capture the graph refers to the optimizer's state tensors BY ADDRESS; torch's load_state_dict
    replaces them and the replays would go on updating the orphans.  It raises from now on and names the
    in-place loader."""
    if getattr(optimizer, '_parl_plain_load_state_dict', None) is not None:
        return
    optimizer._parl_plain_load_state_dict = optimizer.load_state_dict

    def refuse(state_dict):
        raise RuntimeError('this optimizer\'s state is referenced by a captured hipGraph (GraphedLearn): use '
                           'parl_amd.algorithms.impala.graphed.load_optimizer_state_inplace(optimizer, state_dict)')

    optimizer.load_state_dict = refuse


class GraphedLearn(object):
    def __init__(self, alg, B, obs_shape, act_dim, entropy_coeff=-0.01, pool=None):
        self.alg, self.B, self.T = alg, int(B), int(alg.sample_batch_steps)
        self._clip_adam = None   # ops.ClipAdam once the optimizer is capturable (False: not covered)
        dev = next(alg.model.parameters()).device
        assert dev.type == 'cuda', 'GraphedLearn needs the device path (there is no CPU fallback)'
        self.device = dev
        N = self.T * self.B
        self.obs = torch.zeros((N, ) + tuple(obs_shape), dtype=torch.uint8, device=dev)
        self.actions = torch.zeros(N, dtype=torch.int64, device=dev)
        self.behaviour_logits = torch.zeros((N, act_dim), dtype=torch.float32, device=dev)
        self.rewards = torch.zeros(N, dtype=torch.float32, device=dev)
        self.dones = torch.zeros(N, dtype=torch.bool, device=dev)
        # (total, pi, vf, entropy, kl, 1) of the last replay = sums @ M + c; running sums of those
        self.out = torch.zeros(6, dtype=torch.float64, device=dev)
        self.acc = torch.zeros(6, dtype=torch.float64, device=dev)
        self._one = torch.tensor([0, 0, 0, 0, 0, 1], dtype=torch.float64, device=dev)
        self._M = None
        self.entropy_coeff = None
        self.pool = pool
        self.graphs = None
        self.replays = 0
        self.allreduce_in_graph = False
        self.allreduce_capture_error = None
        self._capture(float(entropy_coeff))

    # ---- the captured body -------------------------------------------------------------------
    def _forward_backward(self):
        alg, T, B = self.alg, self.T, self.B
        m = alg.model
        if alg._can_fuse_heads(self.obs, True):
            # heads + loss + the heads' backward are ONE kernel whose outputs ARE gradients: hand them to autograd
            # (trunk) and to the heads' .grad directly instead of building total_loss and multiplying every
            # gradient by d total / d total = 1 (12 single-element launches less per update)
            from ... import ops
            hidden = alg._heads_in_chunks(self.obs, True, trunk_only=True)
            out = ops.impala_heads_loss(hidden.detach().reshape(T, B, 256), m.policy_fc.weight.detach(),
                                        m.policy_fc.bias.detach(), m.value_fc.weight.detach(), m.value_fc.bias.detach(),
                                        self.behaviour_logits.view(T, B, -1), self.actions.view(T, B),
                                        self.rewards.view(T, B), self.dones.view(T, B), alg.gamma, alg.clip_rho_threshold,
                                        alg.clip_pg_rho_threshold, alg.vf_loss_coeff, self.entropy_coeff)
            assert out is not None
            vs, pg, gh, gwp, gbp, gwv, gbv, sums = out
            alg._zero_grad()
            hidden.backward(gh.reshape(T * B, 256))
            for p, g in ((m.policy_fc.weight, gwp), (m.policy_fc.bias, gbp), (m.value_fc.weight, gwv.reshape(1, 256)),
                         (m.value_fc.bias, gbv.reshape(1))):
                if p.grad is None:
                    p.grad = g
                else:  # views of a data-parallel bucket (zeroed by _zero_grad)
                    p.grad.copy_(g)
            if self._M is None:
                self._M = torch.tensor([[1, 1, 0, 0, 0, 0], [alg.vf_loss_coeff, 0, 1, 0, 0, 0],
                                        [self.entropy_coeff, 0, 0, 1, 0, 0], [0, 0, 0, 0, 1.0 / (T * B), 0]],
                                       dtype=torch.float64, device=self.device)
            torch.addmv(self._one, self._M.t(), sums, out=self.out)  # sums = (pi, vf, entropy, kl * N)
        else:
            loss, kl = alg._vtrace_loss(self.obs, self.actions, self.behaviour_logits, self.rewards, self.dones,
                                        self.entropy_coeff, True)
            alg._zero_grad()
            loss.total_loss.backward()
            self.out.copy_(torch.stack([loss.total_loss.detach().double(), loss.pi_loss.detach().double(),
                                        loss.vf_loss.detach().double(), loss.entropy.detach().double(),
                                        kl.detach().double(), self._one[5]]))
        self.acc.add_(self.out)

    def _clip_and_step(self):
        """global-norm clip + Adam: two launches of ops.ClipAdam on the optimizer's own state tensors where it covers
        the optimizer (torch.optim.Adam, one group, <= 16 parameters — every model of this path), else the framework
        pair (~12 launches, one of them 39 us: torch's fused Adam gives each workgroup a 65,536-element chunk)"""
        alg = self.alg
        if self._clip_adam is None and os.environ.get('PARL_AMD_CLIP_ADAM', '1') != '0' and ops.ClipAdam.supported(alg.optimizer):
            self._clip_adam = ops.ClipAdam(alg.optimizer, alg.grad_clip_norm)
        if self._clip_adam:
            self._clip_adam.step()
            return
        self._clip_adam = False
        torch.nn.utils.clip_grad_norm_(alg.model.parameters(), max_norm=alg.grad_clip_norm)
        alg.optimizer.step()

    def _capture(self, entropy_coeff):
        alg, dev = self.alg, self.device
        opt = alg.optimizer
        make_capturable(opt, dev)
        self.entropy_coeff = entropy_coeff
        self._M = None
        params = [p for p in alg.model.parameters()]
        # the warm-up iterations run real updates on the static (zero) inputs: keep the parameters, the
        # optimizer state and the learning rate, put them back afterwards (in place: the graph holds addresses)
        saved_p = [p.detach().clone() for p in params]
        had_state = {p: {k: v.clone() for k, v in opt.state[p].items() if isinstance(v, torch.Tensor)}
                     for p in params if p in opt.state and opt.state[p]}
        lrs = [g['lr'].clone() for g in opt.param_groups]
        for g in opt.param_groups:
            g['lr'].zero_()
        acc0 = self.acc.clone()
        try:
            self._warm_up_and_capture(dev)
        finally:  # also when the capture fails: never leave the optimizer with lr = 0 and warm-up state
            torch.cuda.synchronize(dev)
            with torch.no_grad():
                for p, s in zip(params, saved_p):
                    p.copy_(s)
                for p in params:
                    st = opt.state.get(p, {})
                    for k, v in st.items():
                        if isinstance(v, torch.Tensor):
                            if p in had_state and k in had_state[p]:
                                v.copy_(had_state[p][k])
                            else:
                                v.zero_()
                for g, lr in zip(opt.param_groups, lrs):
                    g['lr'].copy_(lr)
                    g.pop('_lr_value', None)
                self.acc.copy_(acc0)
            torch.cuda.synchronize(dev)

    def _warm_up_and_capture(self, dev):
        alg = self.alg
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(cur)
        split = alg.grad_hook is not None
        with torch.cuda.stream(side):
            for _ in range(2):
                self._forward_backward()
                if split:
                    alg.grad_hook(alg.model)
                self._clip_and_step()
        cur.wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graphs = []
        kw = {'pool': self.pool} if self.pool is not None else {}
        from ... import dist as pdist
        tl = pdist.graph_capture_kwargs()   # (a process group's watchdog thread queries events while we capture)
        self.allreduce_in_graph = False
        if split and self._collective_capturable():
            # Data-parallel update as ONE hipGraph: forward + backward | RCCL all-reduce of the flat bucket | clip +
            # Adam.  RCCL's kernels are capturable (torch's ProcessGroupNCCL enqueues them on its own stream behind
            # events, which a capture records as graph dependencies); the eager collective between two graphs cost a
            # host call, two graph launches and an idle learner stream 51 times per rollout.  thread_local capture
            # mode: the process group's watchdog thread queries events while we capture.  If the capture fails
            # (a backend that cannot be captured), the two-graph form below takes over.
            g1 = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g1, **{**kw, 'capture_error_mode': 'thread_local'}):
                    self._forward_backward()
                    alg.grad_hook(alg.model)
                    self._clip_and_step()
                self.graphs = [g1]
                self.allreduce_in_graph = True
            except Exception as e:   # noqa: BLE001 (any capture failure: fall back, keep the reason)
                self.allreduce_capture_error = '%s: %s' % (type(e).__name__, str(e).split('\n')[0])
                torch.cuda.synchronize(dev)
                g1 = None
        if self.graphs:
            pass
        elif split:
            g1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1, **kw, **tl):
                self._forward_backward()
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, pool=g1.pool(), **tl):
                self._clip_and_step()
            self.graphs = [g1, g2]
        else:
            g1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1, **kw, **tl):
                self._forward_backward()
                self._clip_and_step()
            self.graphs = [g1]
        if self.pool is None:
            self.pool = g1.pool()
        _guard_load_state_dict(alg.optimizer)
        for p in list(alg.model.parameters()) + list(alg.model.buffers()):
            p._parl_graph_written = True  # replays write it without moving its version counter (ops._cached_layout)

    @staticmethod
    def _collective_capturable():
        """the gradient all-reduce can be captured into the update's graph: an RCCL group (gloo moves the bucket
        through the host) and not switched off (PARL_AMD_GRAPH_ALLREDUCE=0: the two-graph form, for A/B runs)"""
        import torch.distributed as dist
        return bool(dist.is_available() and dist.is_initialized() and dist.get_backend() == 'nccl' and
                    os.environ.get('PARL_AMD_GRAPH_ALLREDUCE', '1') != '0')

    # ---- per update --------------------------------------------------------------------------
    def load(self, batch, b0, E):
        """copy sequences [b0, b0+B) of a time-major rollout batch (flat [T*E, ...] tensors, rows ordered
        t-major: DeviceRollout.collect_end) into the static inputs"""
        T, B = self.T, self.B

        def cut(x):
            return x.view((T, E) + tuple(x.shape[1:]))[:, b0:b0 + B]

        if hasattr(batch['obs'], 'gather_sequences'):   # rollout.RingBatch: the stacks straight from the frame ring
            batch['obs'].gather_sequences(b0, B, self.obs)
        else:
            self.obs.view((T, B) + tuple(self.obs.shape[1:])).copy_(cut(batch['obs']))
        self.actions.view(T, B).copy_(cut(batch['actions']))
        self.behaviour_logits.view(T, B, -1).copy_(cut(batch['behaviour_logits']))
        self.rewards.view(T, B).copy_(cut(batch['rewards']))
        self.dones.view(T, B).copy_(cut(batch['dones']))

    def replay(self, learning_rate, entropy_coeff=None):
        """one parameter update on the loaded batch (enqueued on the current stream)"""
        if entropy_coeff is not None and float(entropy_coeff) != self.entropy_coeff:
            self._capture(float(entropy_coeff))
        set_lr(self.alg.optimizer, float(learning_rate))
        self.graphs[0].replay()
        if len(self.graphs) == 2:
            self.alg.grad_hook(self.alg.model)
            self.graphs[1].replay()
        self.replays += 1

    def pop_stats(self):
        """means of (total_loss, pi_loss, vf_loss, entropy, kl) over the replays since the last pop, and
        their number; synchronises the current stream"""
        a = self.acc.tolist()
        self.acc.zero_()
        n = a[5]
        return ([x / n for x in a[:5]] if n else None), int(n)
