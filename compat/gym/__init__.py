"""Minimal `gym` for scripts written against the reference: `gym.make(env_id)` for the Atari ids
the device emulator runs (gym 0.12.1 call sites: examples/IMPALA/actor.py:34,
benchmark/torch/a2c/actor.py:37, train.py:41).  The returned object is a handle that
parl.env.atari_wrappers.wrap_deepmind / parl.env.vector_env.VectorEnv turn into a device-resident
vector of envs; it answers observation_space / action_space / spec.id / unwrapped and refuses to
be stepped on the host.  Put compat/ on PYTHONPATH only when the real gym is not wanted."""
from parl_amd.env.atari_wrappers import DeviceAtariEnv

__version__ = '0.12.1'


def make(env_id, **kwargs):
    if kwargs:
        raise TypeError('gym.make: keyword arguments are not supported on the device path: %r' % (kwargs, ))
    return DeviceAtariEnv(env_id)
