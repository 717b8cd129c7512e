"""parl.env.atari_wrappers for scripts written against the reference
(parl/env/atari_wrappers.py:32-41,44-100,356-385): `wrap_deepmind`, `MonitorEnv`,
`get_wrapper_by_cls`.

The reference wraps ONE gym env in nine Python wrappers and steps it on the host.  Here the whole
chain is one state machine inside the env kernel (csrc/atari_env.hip) and exists only for whole
vectors of envs, so these objects are HANDLES: `gym.make(id)` (compat/gym) gives a `DeviceAtariEnv`
naming the game, `wrap_deepmind` records `dim` / `obs_format` and answers the questions the
reference scripts ask a wrapped env (observation_space.shape, action_space.n, MonitorEnv
statistics), and `parl.env.vector_env.VectorEnv([...handles...])` creates the DeviceVectorEnv
that actually runs them.  A handle cannot be stepped on its own (reset()/step() raise): the
reference never does that on this path (examples/IMPALA/actor.py:33-39,
benchmark/torch/a2c/actor.py:36-42 only hand the list to VectorEnv)."""
import collections

from .device_vector_env import GAMES
from .. import _native as N

__all__ = ['wrap_deepmind', 'MonitorEnv', 'TestEnv', 'get_wrapper_by_cls', 'DeviceAtariEnv']

Box = collections.namedtuple('Box', ['shape', 'dtype', 'low', 'high'])
Discrete = collections.namedtuple('Discrete', ['n'])
Spec = collections.namedtuple('Spec', ['id'])


class DeviceAtariEnv(object):
    """what compat `gym.make('PongNoFrameskip-v4')` returns: the raw env of the chain (a handle)"""

    def __init__(self, env_id):
        if env_id not in GAMES:
            raise ValueError('unsupported env %r (have %s)' % (env_id, sorted(GAMES)))
        self.env_id = env_id
        self.spec = Spec(env_id)
        self.observation_space = Box((210, 160, 3), 'uint8', 0, 255)
        self.action_space = Discrete(N.lib().parlhip_atari_num_actions(GAMES[env_id][1]))
        self.unwrapped = self

    def _no_host_stepping(self, *a, **k):
        raise RuntimeError('a device Atari env is stepped through parl.env.vector_env.VectorEnv (one kernel for '
                           'all envs); single-env host stepping does not exist on this path')

    reset = step = render = _no_host_stepping

    def close(self):
        pass


class MonitorEnv(object):
    """parl/env/atari_wrappers.py:44-100: per-env (unclipped return, length in raw frames) of the closed
    episodes.  Same bookkeeping as the reference: CUMULATIVE `_episode_rewards` / `_episode_lengths` lists
    plus the `_num_returned` cursor of next_episode_results() (:97-100), so get_episode_rewards() /
    get_episode_lengths() keep returning every episode of the run.  Filled by the VectorEnv that owns the
    env: the device reports an episode when it closes; get_total_steps() is the reference's count of EVERY raw
    step (:73-77,90-91) — the closed episodes' lengths plus the running episode's counter, read from the env's
    state blob on the device (`_running_steps`, installed by VectorEnv; one 4-byte D2H per call)."""

    def __init__(self):
        self._episode_rewards = []
        self._episode_lengths = []
        self._num_episodes = 0
        self._num_returned = 0
        self._total_steps = 0       # raw steps of the closed episodes
        self._running_steps = None  # () -> raw steps of the running episode (device counter)

    def _push(self, ret, length):
        self._episode_rewards.append(float(ret))
        self._episode_lengths.append(int(length))
        self._num_episodes += 1
        self._total_steps += int(length)

    def next_episode_results(self):
        for i in range(self._num_returned, len(self._episode_rewards)):
            yield (self._episode_rewards[i], self._episode_lengths[i])
        self._num_returned = len(self._episode_rewards)

    def get_episode_rewards(self):
        return self._episode_rewards

    def get_episode_lengths(self):
        return self._episode_lengths

    def get_total_steps(self):
        return self._total_steps + (int(self._running_steps()) if self._running_steps is not None else 0)


class TestEnv(object):
    """parl/env/atari_wrappers.py:309-353, the evaluation wrapper `wrap_deepmind(test=True)` puts on top of the chain.
    In the reference it passes step() through unchanged (life losses still end a "done" segment) and, in every
    reset(), compares MonitorEnv's episode count with the end of the current evaluation window: once
    `test_episodes` more REAL episodes have closed, get_real_done() turns True and get_eval_rewards() returns their
    unclipped returns, until the next reset.  The chain itself lives in the env kernel; what is left of TestEnv is
    this bookkeeping on the handle's MonitorEnv, run by VectorEnv at the places where the reference's VectorEnv
    calls env.reset() (vector_env.py:34-39, :55-57).  Pinned on the reference class itself
    (tests/golden/wrapper_chain_breakout_42_test.npz)."""
    __test__ = False  # not a pytest class

    def __init__(self, monitor, test_episodes=3):
        self._monitor = monitor
        self._test_episodes = int(test_episodes)
        self._was_real_done = False
        self._eval_rewards = None
        self._end_episode = len(monitor.get_episode_rewards()) + self._test_episodes

    def _on_reset(self):
        """the part of TestEnv.reset behind the inner reset (atari_wrappers.py:336-345)"""
        if self._get_curr_episode() >= self._end_episode:
            self._was_real_done = True
            self._eval_rewards = self._monitor.get_episode_rewards()[-self._test_episodes:]
            self._end_episode = self._end_episode + self._test_episodes
        else:
            self._was_real_done = False
            self._eval_rewards = None

    def get_eval_rewards(self):
        return self._eval_rewards

    def get_real_done(self):
        return self._was_real_done

    def _get_curr_episode(self):
        return len(self._monitor.get_episode_rewards())


class WrappedDeviceAtariEnv(object):
    """result of wrap_deepmind on a device handle"""

    def __init__(self, env, dim, obs_format, test_episodes=None):
        self.env = env
        self.unwrapped = env
        self.env_id = env.env_id
        self.spec = env.spec
        self.dim = int(dim)
        self.obs_format = obs_format
        shape = (4, self.dim, self.dim) if obs_format == 'NCHW' else (self.dim, self.dim, 4)
        self.observation_space = Box(shape, 'uint8', 0, 255)
        self.action_space = env.action_space
        self.monitor = MonitorEnv()
        self.test_env = TestEnv(self.monitor, test_episodes) if test_episodes else None

    reset = step = DeviceAtariEnv._no_host_stepping

    def get_eval_rewards(self):   # the outermost wrapper of a test env is TestEnv: its two getters live here
        if self.test_env is None:
            raise AttributeError('get_eval_rewards: this env was wrapped without test=True')
        return self.test_env.get_eval_rewards()

    def get_real_done(self):
        if self.test_env is None:
            raise AttributeError('get_real_done: this env was wrapped without test=True')
        return self.test_env.get_real_done()

    def close(self):
        pass


def wrap_deepmind(env, dim=84, framestack=True, obs_format='NHWC', test=False, test_episodes=3):
    """same signature as the reference (atari_wrappers.py:356-385)"""
    if not isinstance(env, DeviceAtariEnv):
        raise TypeError('wrap_deepmind: expected the env returned by the device gym.make, got %r' % (env, ))
    if dim not in (42, 84):
        raise ValueError('wrap_deepmind: dim must be 42 or 84 on the device path')
    if not framestack:
        raise ValueError('wrap_deepmind: the device path always stacks 4 frames (the examples do)')
    if test and int(test_episodes) < 1:
        raise ValueError('wrap_deepmind(test=True): test_episodes must be >= 1')
    if obs_format not in ('NHWC', 'NCHW'):
        raise ValueError("obs_format should be one of ['NHWC', 'NCHW']")
    return WrappedDeviceAtariEnv(env, dim, obs_format, test_episodes=int(test_episodes) if test else None)


def get_wrapper_by_cls(env, cls):
    """atari_wrappers.py:32-41: the wrapper of class `cls` in env's chain, or None"""
    if cls is MonitorEnv and isinstance(env, WrappedDeviceAtariEnv):
        return env.monitor
    if cls is TestEnv and isinstance(env, WrappedDeviceAtariEnv):
        return env.test_env
    return None
