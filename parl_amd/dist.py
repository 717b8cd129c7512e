"""One process per GPU over RCCL (torch.distributed backend "nccl" on ROCm) — SURVEY.md §8(e).

The reference has no collective code at all (actors are xparl RPC processes, one learner GPU).
Here envs shard by rank (env ids rank*E .. rank*E+E-1 keep their RNG streams), every rank runs
emulator + policy + V-trace + fwd/bwd on its own shard, and ONE exchange happens per update:
an all-reduce (SUM — the losses are sums, impala.py:67-79, so the data-parallel gradient of the
union batch is the sum) of the flattened gradient, followed by global-norm clipping on the
REDUCED gradient and an identical Adam step on every rank.  Small per-step tensors (actions,
behaviour logits, rewards, dones: 41 B/step at A=6) can be all-gathered for global statistics;
observations are never gathered (1.45 GB per rank at 84x84 would buy nothing under DP)."""
import datetime
import os
import tempfile

import torch
import torch.distributed as dist

_debug_log = None


def collective_log_tail(n=12):
    """the last lines RCCL wrote at NCCL_DEBUG=WARN in this process (init() points NCCL_DEBUG_FILE at a per-process
    file unless the caller set one), for error reports"""
    try:
        with open(_debug_log) as f:
            return [x.rstrip() for x in f.readlines()[-n:]]
    except (OSError, TypeError):
        return []


def describe():
    """what the process group actually is: backend and the world size IT reports (not the env's)"""
    if not active():
        return 'none (single process)'
    return '%s, %d rank(s) in the group' % (dist.get_backend(), dist.get_world_size())


def _rendezvous_store(addr, port, rank, world, timeout):
    """the store `init_method='env://'` would create (torch.distributed.rendezvous._create_c10d_store), with OUR
    timeout: under torchrun the workers are clients of the agent's store (TORCHELASTIC_USE_AGENT_STORE) behind a
    per-attempt prefix; otherwise rank 0 hosts it"""
    if os.environ.get('TORCHELASTIC_USE_AGENT_STORE') == 'True':
        tcp = dist.TCPStore(addr, port, world, False, timeout=timeout)
        return dist.PrefixStore('/worker/attempt_%s' % os.environ.get('TORCHELASTIC_RESTART_COUNT', '0'), tcp)
    return dist.TCPStore(addr, port, world, rank == 0, timeout=timeout, multi_tenant=True)


def init(backend=None, force=False, timeout_s=None, collective_timeout_s=None):
    """Initialise from the torchrun env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  A single
    process creates no group unless force=True (one-rank RCCL group: tests, PARL_AMD_FORCE_DIST=1).
    Two separate limits: `timeout_s` (default PARL_AMD_DIST_TIMEOUT or 120) bounds the RENDEZVOUS — a
    rank that never shows up makes init raise after this long instead of hanging (torch's default is 10-30
    minutes); `collective_timeout_s` (default PARL_AMD_COLLECTIVE_TIMEOUT or 900) is what RCCL's watchdog allows a
    single collective before it aborts the process — long enough for a peer's first-iteration graph captures,
    checkpoint or log I/O, short enough that a dead peer does not hang the job for good."""
    global _debug_log
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    force = force or os.environ.get('PARL_AMD_FORCE_DIST', '').strip().lower() in ('1', 'true', 'yes', 'on')
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            # PARL_AMD_DIST_BACKEND=gloo: test hook (e.g. two ranks sharing the one GPU of a test box)
            backend = os.environ.get('PARL_AMD_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local)
        elif torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        if timeout_s is None:
            timeout_s = float(os.environ.get('PARL_AMD_DIST_TIMEOUT', '120'))
        if collective_timeout_s is None:
            collective_timeout_s = float(os.environ.get('PARL_AMD_COLLECTIVE_TIMEOUT', str(max(900.0, timeout_s))))
        if backend == 'nccl':   # RCCL's own complaints, kept per process for error reports (collective_log_tail)
            os.environ.setdefault('NCCL_DEBUG', 'WARN')
            if 'NCCL_DEBUG_FILE' not in os.environ:
                _debug_log = os.path.join(tempfile.gettempdir(), 'parl_amd_rccl_rank%d_pid%d.log' % (rank, os.getpid()))
                os.environ['NCCL_DEBUG_FILE'] = _debug_log
            else:
                _debug_log = os.environ['NCCL_DEBUG_FILE']
        try:
            # the store carries the rendezvous limit, the group the collective limit
            store = _rendezvous_store(os.environ['MASTER_ADDR'], int(os.environ['MASTER_PORT']), rank, world,
                                      datetime.timedelta(seconds=timeout_s))
            dist.init_process_group(backend=backend, store=store, rank=rank, world_size=world,
                                    timeout=datetime.timedelta(seconds=collective_timeout_s))
        except Exception as e:
            raise RuntimeError('process group rendezvous failed on rank %d of %d (%s at %s:%s, timeout %.0f s): %s: %s' %
                               (rank, world, backend, os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'], timeout_s,
                                type(e).__name__, e))
    return rank, local, world


def graph_capture_kwargs():
    """Keyword arguments for torch.cuda.graph(...) of EVERY capture in this package.  With a process group alive,
    ProcessGroupNCCL's watchdog thread polls its outstanding collectives with hipEventQuery; under the default
    capture mode ('global') any such call from any thread while a capture is open fails with
    hipErrorStreamCaptureUnsupported, the watchdog throws and the process aborts (seen once in 265 runs of the GPU
    suite: an eager all-reduce still outstanding when a rollout segment was being captured).  'thread_local' confines
    the check to the capturing thread, which is the only one of ours that issues HIP work."""
    if dist.is_available() and dist.is_initialized():
        return {'capture_error_mode': 'thread_local'}
    return {}


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def active():
    """collectives run whenever a process group exists — also a one-rank group (a 1-GPU box
    exercising RCCL itself); a plain single process never creates one"""
    return dist.is_available() and dist.is_initialized()


def broadcast_model(model, src=0):
    """init-time weight broadcast so every rank starts from rank 0's parameters"""
    if not active():
        return
    for p in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(p.data, src=src)
    f = getattr(model, 'refresh_actor_layout', None)   # derived copies of the weights (.data writes move no version counter)
    if f is not None:
        f()


class FlatGradAllReduce(object):
    """grad_hook for IMPALA/A2C/PPO: one bucket = the whole model (4.0 MB / 10.9 MB fp32), one
    all-reduce per update.  The flat buffer is allocated once.  The algorithms call
    `hook.zero_grad()` instead of `optimizer.zero_grad(set_to_none=True)` when a hook is installed:
    it zeroes the flat buffer and makes every `p.grad` a view into it, so backward accumulates
    straight into the bucket and there is no pack / unpack copy.  A gradient that is not a view
    (someone reset `.grad`) is packed; a parameter without gradient on this rank contributes
    zeros AND receives the reduced value, so every replica takes the same Adam step."""

    def __init__(self, model, average=False):
        # average=True: divide by the world size after the sum (PPO / any mean-reduced loss: the
        # data-parallel gradient of the union minibatch is the mean of the per-rank gradients)
        self.average = average
        self.params = [p for p in model.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=self.params[0].dtype, device=self.params[0].device)
        self._shared = None   # SharedDeviceAllReduce when the ranks share one GPU (decided at the first call)
        off = 0
        self.views = []
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def zero_grad(self):
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            p.grad = v

    def __call__(self, model=None):
        if not active():
            return
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
                p.grad = v
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
                p.grad = v
        if self._shared is None:
            self._shared = SharedDeviceAllReduce.create(self.flat) or False
        if self._shared:
            self._shared(self.flat)
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        if self.average:
            self.flat.div_(world_size())


class SharedDeviceAllReduce(object):
    """SUM all-reduce for ranks that SHARE one GPU (PARL_AMD_SHARE_GPU: a test box with fewer devices than ranks, where
    RCCL cannot be used — it refuses two ranks on one device — and gloo stages every 4 MB bucket through the host:
    ~25 updates/s, too few to see a data-parallel run learn).  Every rank owns two device slots of the bucket's size and
    opens its peers' slots through HIP IPC (torch's CUDA tensor sharing; dmabuf IPC: HSA_ENABLE_IPC_MODE_LEGACY=0).
    One call: copy the bucket into my slot of this call's parity -> wait for the copy on the host -> gloo barrier
    (every rank's slot is complete) -> bucket = slot of rank 0 + slot of rank 1 + ... in rank order on the device
    (the same order on every rank: bit-identical replicas).  The other parity is written by the next call, and a slot
    is rewritten only two barriers later, when every peer's sum of it has long been waited for.  Functional
    infrastructure for one-GPU tests of the data-parallel path, not a scaling path (one host wait per update)."""

    @staticmethod
    def create(flat):
        if not (active() and os.environ.get('PARL_AMD_SHARE_GPU') and flat.is_cuda and dist.get_backend() == 'gloo'
                and os.environ.get('PARL_AMD_SHARED_ALLREDUCE', '1') != '0'):
            return None
        try:
            return SharedDeviceAllReduce(flat)
        except Exception as e:   # noqa: BLE001 (IPC not available: gloo's own all-reduce takes over, all ranks alike)
            import warnings
            warnings.warn('shared-device all-reduce unavailable (%s: %s); using gloo' % (type(e).__name__, e))
            return None

    def __init__(self, flat):
        from torch.multiprocessing.reductions import reduce_tensor
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.mine = torch.zeros((2, flat.numel()), dtype=flat.dtype, device=flat.device)
        torch.cuda.synchronize(flat.device)
        handles = [None] * self.world
        ok = True
        try:
            mine = reduce_tensor(self.mine)
        except Exception:   # noqa: BLE001
            mine, ok = None, False
        dist.all_gather_object(handles, mine)
        if not ok or any(h is None for h in handles):   # every rank takes the same decision
            raise RuntimeError('a rank could not export its slots')
        self.slots = [self.mine if r == self.rank else handles[r][0](*handles[r][1]) for r in range(self.world)]
        self.parity = 0
        dist.barrier()

    def __call__(self, flat):
        p = self.parity
        self.mine[p].copy_(flat)
        torch.cuda.current_stream(flat.device).synchronize()
        dist.barrier()
        flat.copy_(self.slots[0][p])
        for r in range(1, self.world):
            flat.add_(self.slots[r][p])
        self.parity = p ^ 1


_gather_bufs = {}


def all_gather_small(tensors, slot=0):
    """All-gather a dict of small per-step tensors along a new leading rank dim.  The [world, ...] receive
    buffers are allocated once per (slot, key, shape, dtype, device) and reused: one call per learner update
    and slot.  The result of a call is valid until the next call with the same slot and keys — callers that
    keep several results alive at once (one per env group) pass a different `slot` for each."""
    w = world_size()
    if not active():
        return {k: v.unsqueeze(0) for k, v in tensors.items()}
    out = {}
    for k, v in tensors.items():
        v = v.contiguous()
        key = (slot, k, w, tuple(v.shape), v.dtype, v.device)
        buf = _gather_bufs.get(key)
        if buf is None:
            buf = _gather_bufs[key] = torch.empty((w, ) + tuple(v.shape), dtype=v.dtype, device=v.device)
        if v.is_cuda and dist.get_backend() == 'nccl':
            dist.all_gather_into_tensor(buf, v)
        else:
            parts = [torch.empty_like(v) for _ in range(w)]
            dist.all_gather(parts, v)
            buf = torch.stack(parts)
        out[k] = buf
    return out


def all_reduce_max_scalar(x):
    """max over ranks of a python float (bench timing)"""
    if not active():
        return x
    dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_gather_scalar(x):
    """[x of rank 0, x of rank 1, ...] for a python float (bench.py: per-rank timings on rank 0's line)"""
    if not active():
        return [float(x)]
    dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    t = torch.zeros(world_size(), dtype=torch.float64, device=dev)
    t[dist.get_rank()] = float(x)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t.tolist()]


def time_allreduce(hook, stream=None, iters=30):
    """the gradient all-reduce of a FlatGradAllReduce bucket by itself: median / min / max microseconds of `iters`
    event-timed calls on `stream` (every rank calls this at the same point; the collective is the real one on the
    real bucket, whose contents are garbage afterwards — call it outside an update)"""
    if not active() or hook is None:
        return None
    st = stream or torch.cuda.current_stream(hook.flat.device)
    ts = []
    with torch.cuda.stream(st):
        for _ in range(3):
            hook.flat.zero_()
            dist.all_reduce(hook.flat, op=dist.ReduceOp.SUM)
        st.synchronize()
        evs = []
        for _ in range(iters):
            hook.flat.zero_()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st)
            dist.all_reduce(hook.flat, op=dist.ReduceOp.SUM)
            b.record(st)
            evs.append((a, b))
        st.synchronize()
        ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    return {'n': len(ts), 'median_us': ts[len(ts) // 2], 'min_us': ts[0], 'max_us': ts[-1],
            'bytes': hook.flat.numel() * hook.flat.element_size()}


def barrier():
    if active():
        dist.barrier()
