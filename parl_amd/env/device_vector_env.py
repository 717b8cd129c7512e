"""DeviceVectorEnv — VectorEnv([wrap_deepmind(gym.make(id), dim, obs_format='NCHW')] * E) with
every env resident on one MI355X (one env per wavefront pair: the 6507 on one wave, its picture on a second).

Mirrors the contract of parl/env/vector_env.py:26-63 (``reset() -> obs batch``,
``step(actions) -> (obs, rewards, dones, infos)`` with auto-reset: the obs returned for a done
env is its reset obs while reward/done describe the terminal transition) and of
parl/env/atari_wrappers.py:356-385 (wrapper order).  Differences are the container types:
tensors on the GPU instead of Python lists of numpy arrays; `VectorEnvAdapter` below gives the
list-of-numpy view the reference's actor.py expects.

All compute goes through the C ABI (include/parl_hip.h); there is no CPU fallback.
"""
import hashlib
import os

import numpy as np
import torch

from .. import _native as N
from .. import ops

_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(os.path.dirname(_HERE))

# gym id -> (rom name, C-ABI game id, md5 of the cartridge ALE ships for that id)
GAMES = {
    'PongNoFrameskip-v4': ('pong', 1, '60e0ea3cbe0913d39803477945e9e5ec'),
    'BreakoutNoFrameskip-v4': ('breakout', 2, 'f34f08e5eb96e500e851a80be3277a56'),
}


def find_rom(name, rom_dir=None):
    """Cartridge bytes for `name` ('pong', 'breakout').  ROMs are user-supplied data (as with
    ALE): looked up in rom_dir, $PARL_AMD_ROM_DIR, <repo>/roms."""
    dirs = [rom_dir, os.environ.get('PARL_AMD_ROM_DIR'), os.path.join(_REPO, 'roms')]
    for d in dirs:
        if d and os.path.exists(os.path.join(d, name + '.bin')):
            return open(os.path.join(d, name + '.bin'), 'rb').read()
    raise FileNotFoundError(
        'cartridge %s.bin not found (looked in %s). Put the ROM there or set PARL_AMD_ROM_DIR.' %
        (name, [d for d in dirs if d]))


class DeviceVectorEnv(object):
    def __init__(self, env_name, num_envs, dim=84, horizon=1, seed=0, env_id0=0, device=None,
                 rom_dir=None, max_episode_steps=400000, use_reset_cache=True, rom_bytes=None, native=True):
        if env_name not in GAMES:
            raise ValueError('unsupported env %r (have %s)' % (env_name, sorted(GAMES)))
        name, self.game, md5 = GAMES[env_name]
        rom = rom_bytes if rom_bytes is not None else find_rom(name, rom_dir)
        if hashlib.md5(rom).hexdigest() != md5:
            raise ValueError('%s.bin md5 mismatch: not the cartridge ALE uses for %s' % (name, env_name))
        self.env_name, self.envs_num, self.dim = env_name, int(num_envs), int(dim)
        self.horizon = int(horizon)
        self.seed, self.env_id0 = int(seed), int(env_id0)
        self.max_episode_steps = int(max_episode_steps)
        self.device = torch.device(device if device is not None else 'cuda')
        L = N.lib()
        self.act_dim = L.parlhip_atari_num_actions(self.game)
        self.obs_shape = (4, self.dim, self.dim)
        E, dev = self.envs_num, self.device
        u8 = dict(dtype=torch.uint8, device=dev)
        # --- cartridge: pre-decoded on the host, resident on the device
        self.rom_size = len(rom)
        table = np.zeros(self.rom_size, np.uint32)
        rom_np = np.frombuffer(rom, np.uint8)
        N.check(L.parlhip_atari_rom_table_build(rom_np.ctypes.data, self.rom_size, table.ctypes.data),
                'parlhip_atari_rom_table_build')
        if not native:  # A/B and tests: run the 6507 interpreter even if translated code exists
            table[0] &= 0x0fffffff
        # True when the env kernel runs this cartridge as translated native code (else interpreted)
        self.native = int(table[0] >> 28) == self.game
        self.rom_table = torch.from_numpy(table.view(np.int32)).to(dev)
        # --- per-env state, raw frame pairs, per-step outputs
        self.states = torch.zeros(E * L.parlhip_atari_state_bytes(), **u8)
        self.raw_frames = torch.zeros((E, 2, 210, 160), **u8)
        self.rewards = torch.zeros(E, dtype=torch.float32, device=dev)
        self.dones = torch.zeros(E, **u8)
        self.obs_flags = torch.zeros(E, **u8)
        self.ep_returns = torch.zeros(E, dtype=torch.float32, device=dev)
        self.ep_lengths = torch.zeros(E, dtype=torch.int32, device=dev)
        self.jam = torch.zeros(1, dtype=torch.int32, device=dev)
        # --- frame_post tables
        nb = L.parlhip_frame_post_tables_bytes(self.dim)
        blob = np.zeros(nb, np.uint8)
        N.check(L.parlhip_frame_post_tables_init(blob.ctypes.data, self.dim), 'frame_post_tables_init')
        self.fp_tables = torch.from_numpy(blob).to(dev)
        # --- rollout ring of SINGLE frames: slot t+3 holds the newest frame of obs time t
        self.fsz = self.dim * self.dim
        self.slots = self.horizon + 4
        self.ring = torch.zeros((self.slots, E, self.fsz), **u8)
        self.since = torch.zeros((self.slots, E), **u8)
        # more than one ring (ensure_rings): successive rollouts alternate between them, so that a learner can still
        # read the frames of rollout i-1 IN PLACE while rollout i is written (parl_amd.rollout.RingBatch) — no
        # [T*E, 4, d, d] batch of stacks is materialised per rollout
        self._rings = [(self.ring, self.since)]
        self.ring_index = 0
        self.t = 0
        self.link = None      # elastic launches: [slots, E] slot of the env's previous observation
        self.cur_slot = None  # elastic launches: [E] slot of the env's current observation
        self._env_idx = torch.arange(E, dtype=torch.int32, device=dev)
        self._slot_const = {}
        # VectorEnv.step as ONE launch (parlhip_atari_vec_step_obs) where the kernel supports it: 42 / 84 frames of a
        # 2K cartridge.  PARL_AMD_FUSED_OBS=0: the two-launch form (emulator, then frame_post) — A/B and parity tests
        self.fused_obs = (self.dim in (42, 84) and self.rom_size == 2048 and
                          os.environ.get('PARL_AMD_FUSED_OBS', '1') != '0')
        # --- O(1) real resets
        self.reset_cache = None
        if use_reset_cache:
            self.reset_cache = torch.zeros(L.parlhip_atari_reset_cache_bytes(), **u8)
            N.check(
                L.parlhip_atari_reset_cache_build(
                    N.ptr(self.rom_table), self.rom_size, self.game, self.max_episode_steps,
                    N.ptr(self.reset_cache), N.ptr(self.jam), N.stream_ptr()), 'reset_cache_build')

    # ---------------------------------------------------------------- internals
    def _frame_post(self, slot, ep_acc=None):
        # max-2 + gray + INTER_AREA into ring[slot] and the FrameStack `since` counters, one launch; with
        # ep_acc (f64 [3] on the device) also the MonitorEnv statistics of accumulate_episode_stats
        L = N.lib()
        prev = self.since[slot - 1] if slot > 0 else None
        if ep_acc is not None:
            if ep_acc.dtype != torch.float64 or ep_acc.numel() != 3 or ep_acc.device.type != self.device.type:
                raise N.ParlHipError('ep_acc must be float64 [3] on the env device')
            N.check(
                L.parlhip_frame_post_step_u8(
                    N.ptr(self.raw_frames), self.raw_frames.data_ptr() + 210 * 160, 2 * 210 * 160, 1,
                    N.ptr(self.obs_flags), N.ptr(self.ring[slot]), self.fsz, self.envs_num, self.dim,
                    N.ptr(self.fp_tables), N.ptr(prev) if prev is not None else None, N.ptr(self.since[slot]),
                    N.ptr(self.ep_returns), N.ptr(self.ep_lengths), N.ptr(ep_acc), N.stream_ptr()),
                'parlhip_frame_post_step_u8')
            return
        N.check(
            L.parlhip_frame_post_since_u8(
                N.ptr(self.raw_frames), self.raw_frames.data_ptr() + 210 * 160, 2 * 210 * 160, 1,
                N.ptr(self.obs_flags), N.ptr(self.ring[slot]), self.fsz, self.envs_num, self.dim,
                N.ptr(self.fp_tables), N.ptr(prev) if prev is not None else None, N.ptr(self.since[slot]),
                N.stream_ptr()), 'parlhip_frame_post_since_u8')

    def ensure_rings(self, n):
        """allocate rings up to `n` (before or after reset(); the current ring stays current)"""
        if self.link is not None and n > 1:
            raise N.ParlHipError('the elastic ring layout is one circular ring')
        while len(self._rings) < int(n):
            self._rings.append((torch.zeros_like(self.ring), torch.zeros_like(self.since)))

    def gather(self, slots, envs, out=None, ring_index=None):
        """Stacked obs u8 [n,4,dim,dim] for (ring slot, env) pairs (int32 tensors); ring_index: of which ring
        (default: the current one)."""
        n = slots.numel()
        if out is None:
            out = torch.empty((n, 4, self.dim, self.dim), dtype=torch.uint8, device=self.device)
        ring, since = self._rings[self.ring_index if ring_index is None else ring_index]
        N.check(
            N.lib().parlhip_stack_gather_ring_u8(
                N.ptr(ring), N.ptr(since), N.ptr(self.link) if self.link is not None else None, self.slots,
                self.envs_num, self.fsz,
                N.ptr(slots.contiguous()),
                N.ptr(envs.contiguous()), n, N.ptr(out), N.stream_ptr()), 'parlhip_stack_gather_ring_u8')
        return out

    def current_obs(self, out=None):
        return self._obs_at(self.t, out)

    def _obs_at(self, t, out=None):
        """the stacked observation whose newest frame lies in ring slot t + 3"""
        slots = self._slot_const.get(t)
        if slots is None:  # one constant index tensor per ring position, made once
            slots = torch.full((self.envs_num, ), t + 3, dtype=torch.int32, device=self.device)
            self._slot_const[t] = slots
        return self.gather(slots, self._env_idx, out)

    def current_obs_ref(self, out=None):
        """the current observation for a consumer that can read the ring in place (ops.RingObservation; its
        materialize() is current_obs(out)); with elastic launches (gaps in the ring: `link`) the materialised stack"""
        if self.link is not None:
            return self.current_obs(out)
        t = self.t   # the reference names THIS position: materialising it after further steps still gives this stack
        return ops.RingObservation(self.ring, self.since, t + 3, self.dim,
                                   lambda o=None: self._obs_at(t, o if o is not None else out))

    def accumulate_episode_stats(self, acc3):
        """acc3 (f64 [3] on the device) += (episodes closed by the last step, their unclipped
        returns, their lengths) — MonitorEnv.next_episode_results without a host round trip."""
        N.check(
            N.lib().parlhip_episode_stats_accum_f64(N.ptr(self.ep_returns), N.ptr(self.ep_lengths), self.envs_num,
                                                    N.ptr(acc3), N.stream_ptr()), 'parlhip_episode_stats_accum_f64')

    # ---------------------------------------------------------------- VectorEnv contract
    def reset(self):
        """VectorEnv.reset (vector_env.py:34-39) -> obs u8 [E,4,dim,dim] on the device."""
        L = N.lib()
        N.check(
            L.parlhip_atari_vec_reset(
                N.ptr(self.states), N.ptr(self.rom_table), self.rom_size, self.game, N.ptr(self.raw_frames),
                N.ptr(self.obs_flags), self.envs_num, self.seed, self.env_id0, self.max_episode_steps,
                N.ptr(self.jam), N.stream_ptr()), 'parlhip_atari_vec_reset')
        self.t = 0
        self._frame_post(3)
        return self.current_obs()

    def step_async(self, actions, rewards_out=None, dones_out=None, ep_acc=None):
        """Enqueue one VectorEnv.step; results land in self.rewards/dones (or the given [E] slabs
        of a rollout buffer) and the new frame in ring slot t+4.  ep_acc: accumulate_episode_stats(ep_acc)
        folded into the same launches."""
        if actions.dtype != torch.int64:
            raise N.ParlHipError('actions must be int64')
        if self.t >= self.horizon:
            raise N.ParlHipError('rollout ring full: call roll() every `horizon` steps')
        L = N.lib()
        if self.fused_obs:
            # the observation leaves the launch that emulated it (the env's two waves convert its frame pair at
            # their tail): no frame_post launch behind the emulator on the actors' critical path
            if ep_acc is not None and (ep_acc.dtype != torch.float64 or ep_acc.numel() != 3 or
                                       ep_acc.device.type != self.device.type):
                raise N.ParlHipError('ep_acc must be float64 [3] on the env device')
            slot = self.t + 4
            N.check(
                L.parlhip_atari_vec_step_obs(
                    N.ptr(self.states), N.ptr(self.rom_table), self.rom_size, self.game, N.ptr(actions.contiguous()),
                    N.ptr(self.raw_frames), N.ptr(self.rewards if rewards_out is None else rewards_out),
                    N.ptr(self.dones if dones_out is None else dones_out), N.ptr(self.obs_flags),
                    N.ptr(self.ep_returns), N.ptr(self.ep_lengths), self.envs_num, self.seed, self.env_id0,
                    self.max_episode_steps, N.ptr(self.reset_cache), N.ptr(self.jam), N.ptr(self.ring[slot]),
                    self.dim, N.ptr(self.fp_tables), N.ptr(self.since[slot - 1]), N.ptr(self.since[slot]),
                    N.ptr(ep_acc) if ep_acc is not None else None, N.stream_ptr()), 'parlhip_atari_vec_step_obs')
            self.t += 1
            return
        N.check(
            L.parlhip_atari_vec_step(
                N.ptr(self.states), N.ptr(self.rom_table), self.rom_size, self.game, N.ptr(actions.contiguous()),
                N.ptr(self.raw_frames), N.ptr(self.rewards if rewards_out is None else rewards_out),
                N.ptr(self.dones if dones_out is None else dones_out), N.ptr(self.obs_flags),
                N.ptr(self.ep_returns), N.ptr(self.ep_lengths), self.envs_num, self.seed, self.env_id0,
                self.max_episode_steps, N.ptr(self.reset_cache), N.ptr(self.jam), N.stream_ptr()),
            'parlhip_atari_vec_step')
        self.t += 1
        self._frame_post(self.t + 3, ep_acc)

    def can_step_policy(self, hidden_units, act_dim):
        """whether step_policy_async exists for this env and a policy head of that shape"""
        return bool(self.fused_obs and self.link is None and hidden_units == 256 and act_dim == self.act_dim and
                    act_dim <= 6)

    def step_policy_async(self, hidden, w_policy, b_policy, logits_out, actions_out, sample_seed, offset, row0,
                          offset_base=None, rewards_out=None, dones_out=None, ep_acc=None):
        """The actors' step from the trunk output on: policy_fc + the draw (what ops.policy_head_sample_into does —
        same logits, same actions), VectorEnv.step and the observation, ONE launch
        (parlhip_atari_vec_step_policy_obs).  hidden f32 [E, 256]; logits_out [E, A] / actions_out int64 [E]: this
        step's rows of the rollout slabs; the draw of env e is the Philox uniform of (sample_seed; offset
        (+ offset_base[0] on the device), row0 + e)."""
        if self.t >= self.horizon:
            raise N.ParlHipError('rollout ring full: call roll() every `horizon` steps')
        E, A = self.envs_num, self.act_dim
        if not self.can_step_policy(hidden.shape[-1], w_policy.shape[0]):
            raise N.ParlHipError('step_policy_async: not available for this env / head (can_step_policy)')
        if (hidden.dtype != torch.float32 or tuple(hidden.shape) != (E, 256) or logits_out.dtype != torch.float32 or
                tuple(logits_out.shape) != (E, A) or actions_out.dtype != torch.int64 or actions_out.numel() != E):
            raise N.ParlHipError('step_policy_async: hidden f32 [E,256], logits_out f32 [E,A], actions_out int64 [E]')
        if ep_acc is not None and (ep_acc.dtype != torch.float64 or ep_acc.numel() != 3 or
                                   ep_acc.device.type != self.device.type):
            raise N.ParlHipError('ep_acc must be float64 [3] on the env device')
        if offset_base is not None and (offset_base.dtype != torch.int64 or offset_base.numel() != 1 or
                                        not offset_base.is_cuda):
            raise N.ParlHipError('offset_base must be an int64 [1] device tensor')
        wp, bp = w_policy.detach(), b_policy.detach()
        if wp.dtype != torch.float32 or bp.dtype != torch.float32:
            raise N.ParlHipError('step_policy_async: float32 head')
        slot = self.t + 4
        m64 = 2**64 - 1
        N.check(
            N.lib().parlhip_atari_vec_step_policy_obs(
                N.ptr(self.states), N.ptr(self.rom_table), self.rom_size, self.game, N.ptr(self.raw_frames),
                N.ptr(self.rewards if rewards_out is None else rewards_out),
                N.ptr(self.dones if dones_out is None else dones_out), N.ptr(self.obs_flags), N.ptr(self.ep_returns),
                N.ptr(self.ep_lengths), E, self.seed, self.env_id0, self.max_episode_steps, N.ptr(self.reset_cache),
                N.ptr(self.jam), N.ptr(self.ring[slot]), self.dim, N.ptr(self.fp_tables), N.ptr(self.since[slot - 1]),
                N.ptr(self.since[slot]), N.ptr(ep_acc) if ep_acc is not None else None, N.ptr(hidden.contiguous()),
                N.ptr(wp.contiguous()), N.ptr(bp.contiguous()), N.ptr(logits_out), N.ptr(actions_out), 256, A,
                int(sample_seed) & m64, N.ptr(offset_base) if offset_base is not None else None, int(offset) & m64,
                int(row0) & m64, N.stream_ptr()), 'parlhip_atari_vec_step_policy_obs')
        self.t += 1

    # ---------------------------------------------------------------- elastic launches (circular ring)
    def elastic_begin(self):
        """switch the ring to the elastic layout (after reset(): every env's observation is in slot 3)"""
        E = self.envs_num
        self.link = torch.zeros((self.slots, E), dtype=torch.int32, device=self.device)
        self.cur_slot = torch.full((E, ), 3, dtype=torch.int32, device=self.device)

    def elastic_obs(self, out=None):
        """stacked obs every env currently holds (its own ring slot: envs that sat launches out lag behind)"""
        return self.gather(self.cur_slot, self._env_idx, out)

    def step_elastic_async(self, actions, launch, rows_limit, rows_ring, batch_rows, rows_done, row_launch, row_slot,
                           finished, rewards_rows, dones_rows, frame_budget=4):
        """Enqueue one ELASTIC launch (parlhip_atari_vec_step_elastic): every env emulates at most
        `frame_budget` frames; an env inside a life-loss / slow reset sequence goes on with it instead of
        taking `actions[e]`, an env with rows_done >= rows_limit waits.  `launch` counts from the
        reset of the run; the ring is circular: the observation of an env that completed its step lands
        in slot (launch + 4) % slots (untouched otherwise); cur_slot / link / since follow (elastic_begin)."""
        if actions.dtype != torch.int64:
            raise N.ParlHipError('actions must be int64')
        if not hasattr(self, '_ctl'):
            self._ctl = torch.zeros(self.envs_num, dtype=torch.uint8, device=self.device)
        if self.link is None:
            raise N.ParlHipError('call elastic_begin() after reset()')
        slot = (launch + 4) % self.slots
        L = N.lib()
        args = (N.ptr(self.states), N.ptr(self.rom_table), self.rom_size, self.game, N.ptr(actions.contiguous()),
                N.ptr(self.raw_frames), N.ptr(self.rewards), N.ptr(self.dones), N.ptr(self.obs_flags),
                N.ptr(self.ep_returns), N.ptr(self.ep_lengths), self.envs_num, self.seed, self.env_id0,
                self.max_episode_steps, N.ptr(self.reset_cache), N.ptr(self.jam), int(frame_budget), int(launch),
                int(rows_limit), int(rows_ring), int(batch_rows), N.ptr(rows_done), N.ptr(row_launch),
                N.ptr(row_slot), N.ptr(self._ctl), N.ptr(finished), N.ptr(rewards_rows), N.ptr(dones_rows), slot,
                N.ptr(self.cur_slot), N.ptr(self.link), N.ptr(self.since))
        if self.fused_obs:   # the observation of the envs that completed a step, made at the tail of the same launch
            N.check(L.parlhip_atari_vec_step_elastic_obs(*args, N.ptr(self.ring[slot]), self.dim, N.ptr(self.fp_tables),
                                                         N.stream_ptr()), 'parlhip_atari_vec_step_elastic_obs')
            return
        N.check(L.parlhip_atari_vec_step_elastic(*args, N.stream_ptr()), 'parlhip_atari_vec_step_elastic')
        self._frame_post_elastic(slot)

    def _frame_post_elastic(self, slot):
        # max-2 + gray + INTER_AREA into ring[slot] for the envs that completed a step (obs_flags bit 2 clear);
        # the FrameStack bookkeeping of the elastic path is done by the step call itself
        N.check(
            N.lib().parlhip_frame_post_u8(
                N.ptr(self.raw_frames), self.raw_frames.data_ptr() + 210 * 160, 2 * 210 * 160, 1,
                N.ptr(self.obs_flags), N.ptr(self.ring[slot]), self.fsz, self.envs_num, self.dim,
                N.ptr(self.fp_tables), N.stream_ptr()), 'parlhip_frame_post_u8')

    def roll(self):
        """Start the next rollout: the last 4 frame slots become the history of obs time 0 — of the NEXT ring when
        there are several (the ring just written keeps the rollout's frames for whoever still reads them)."""
        T = self.t
        if len(self._rings) > 1:
            nxt = (self.ring_index + 1) % len(self._rings)
            ring, since = self._rings[nxt]
            ring[0:4].copy_(self.ring[T:T + 4])
            since[0:4].copy_(self.since[T:T + 4])
            self.ring, self.since, self.ring_index = ring, since, nxt
        elif T > 0:
            self.ring[0:4].copy_(self.ring[T:T + 4].clone())
            self.since[0:4].copy_(self.since[T:T + 4].clone())
        self.t = 0

    def step(self, actions):
        """VectorEnv.step (vector_env.py:41-63) with tensors: (obs, rewards, dones, info)."""
        if self.t >= self.horizon:
            self.roll()
        self.step_async(actions)
        info = {'episode_returns': self.ep_returns, 'episode_lengths': self.ep_lengths}
        return self.current_obs(), self.rewards, self.dones.bool(), info

    # MonitorEnv's step counter of the RUNNING episode (csrc/atari_defs.hpp: int32 scalar slot S_NUM_STEPS of an env's
    # state blob; tests/test_capi_symbols.py checks both constants against the header)
    _STATE_SCALARS_OFFSET, _SLOT_NUM_STEPS = 192, 32

    def running_episode_steps(self):
        """int32 [E]: raw (per emulated frame) steps of every env's unfinished episode — what MonitorEnv.step has
        added to `_num_steps` since the last real reset (parl/env/atari_wrappers.py:73-77)"""
        blob = self.states.view(self.envs_num, -1)
        o = self._STATE_SCALARS_OFFSET + 4 * self._SLOT_NUM_STEPS
        return blob[:, o:o + 4].contiguous().view(torch.int32).view(-1)

    # ---------------------------------------------------------------- checkpoint
    _STATE_TENSORS = ('states', 'raw_frames', 'ring', 'since', 'rewards', 'dones', 'obs_flags', 'ep_returns',
                      'ep_lengths', 'jam')

    def state_dict(self):
        """Everything that determines the future of the envs: the per-env machine + wrapper state
        blobs (6507 / TIA / RIOT, ALE paddle, lives, noop / reset counters = the Philox offsets of
        the reset stream), the raw frame pair MaxAndSkip still needs, the frame-stack ring with its
        `since` counters and the ring position.  Host tensors; pairs with Agent.save
        (parl/core/torch/agent.py:100-124 saves the model only — a GPU-resident env has no other
        way to survive a restart).  The state may have been written on another HIP stream than the caller's
        (AsyncActorLearner steps the envs on its actor streams, launches ahead of the host): the whole
        device is synchronised first, so the snapshot is never torn."""
        if self.device.type == 'cuda':
            torch.cuda.synchronize(self.device)
        d = {k: getattr(self, k).detach().cpu().clone() for k in self._STATE_TENSORS}
        if len(self._rings) > 1:   # every ring (a collected, not yet learned rollout lives in a non-current one)
            d['rings'] = [(r.detach().cpu().clone(), s_.detach().cpu().clone()) for r, s_ in self._rings]
            d['ring_index'] = self.ring_index
        if self.link is not None:  # elastic ring layout
            d['link'], d['cur_slot'] = self.link.detach().cpu().clone(), self.cur_slot.detach().cpu().clone()
        d['meta'] = {'env_name': self.env_name, 'envs_num': self.envs_num, 'dim': self.dim, 'horizon': self.horizon,
                     'seed': self.seed, 'env_id0': self.env_id0, 'max_episode_steps': self.max_episode_steps,
                     't': self.t}
        return d

    def load_state_dict(self, d):
        m = d['meta']
        for k in ('env_name', 'envs_num', 'dim', 'horizon', 'seed', 'env_id0', 'max_episode_steps'):
            if m[k] != getattr(self, k):
                raise ValueError('DeviceVectorEnv.load_state_dict: %s is %r here but %r in the checkpoint' %
                                 (k, getattr(self, k), m[k]))
        if self.device.type == 'cuda':
            torch.cuda.synchronize(self.device)  # nothing in flight on any stream may still read / write the state
        if 'rings' in d:
            self.ensure_rings(len(d['rings']))
            for (r, s_), (cr, cs) in zip(self._rings, d['rings']):
                r.copy_(cr.to(self.device))
                s_.copy_(cs.to(self.device))
            self.ring_index = int(d['ring_index'])
            self.ring, self.since = self._rings[self.ring_index]
        for k in self._STATE_TENSORS:
            getattr(self, k).copy_(d[k].to(self.device))
        if 'link' in d:
            self.elastic_begin()
            self.link.copy_(d['link'].to(self.device))
            self.cur_slot.copy_(d['cur_slot'].to(self.device))
        self.t = int(m['t'])
        if self.device.type == 'cuda':
            torch.cuda.synchronize(self.device)  # visible to every stream that steps the envs next

    def check_faults(self):
        """Raise if any env hit an emulator fault (undocumented opcode, ...). Synchronises."""
        j = int(self.jam.item())
        if j:
            raise N.ParlHipError('atari emulator fault bits 0x%x' % j)


class VectorEnvAdapter(object):
    """List-of-numpy view with the exact VectorEnv return types (vector_env.py:41-63), for code
    written against the reference (e.g. examples/IMPALA/actor.py:54-91).  Costs a D2H copy per
    step — the device-native loop in parl_amd.rollout does not use it."""

    def __init__(self, env):
        self.env = env
        self.envs_num = env.envs_num

    def reset(self):
        return list(self.env.reset().cpu().numpy())

    def step(self, actions):
        a = torch.as_tensor(np.asarray(actions), dtype=torch.int64, device=self.env.device)
        obs, rew, done, info = self.env.step(a)
        ret = info['episode_returns'].cpu().numpy()
        ln = info['episode_lengths'].cpu().numpy()
        infos = [{'episode': {'r': float(ret[i]), 'l': int(ln[i])}} if ln[i] > 0 else {} for i in range(self.envs_num)]
        return (list(obs.cpu().numpy()), [float(x) for x in rew.cpu().numpy()], [bool(x) for x in done.cpu().numpy()],
                infos)
