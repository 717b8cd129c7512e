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

__all__ = ['wrap_deepmind', 'MonitorEnv', 'get_wrapper_by_cls', 'DeviceAtariEnv']

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
    env: the device reports an episode when it closes, so get_total_steps() counts the frames of closed
    episodes (the reference also counts the running episode's frames, which live on the device here)."""

    def __init__(self):
        self._episode_rewards = []
        self._episode_lengths = []
        self._num_episodes = 0
        self._num_returned = 0
        self._total_steps = 0

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
        return self._total_steps


class WrappedDeviceAtariEnv(object):
    """result of wrap_deepmind on a device handle"""

    def __init__(self, env, dim, obs_format):
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

    reset = step = DeviceAtariEnv._no_host_stepping

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
    if test:
        raise ValueError('wrap_deepmind(test=True): the TestEnv evaluation wrapper is not on the device path')
    if obs_format not in ('NHWC', 'NCHW'):
        raise ValueError("obs_format should be one of ['NHWC', 'NCHW']")
    return WrappedDeviceAtariEnv(env, dim, obs_format)


def get_wrapper_by_cls(env, cls):
    """atari_wrappers.py:32-41: the wrapper of class `cls` in env's chain, or None"""
    if cls is MonitorEnv and isinstance(env, WrappedDeviceAtariEnv):
        return env.monitor
    return None
