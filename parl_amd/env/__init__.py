"""parl_amd.env — on-device counterparts of parl.env (vector_env.py, atari_wrappers.py)."""
from .device_vector_env import DeviceVectorEnv, find_rom, GAMES  # noqa: F401
from .vec_normalize import DeviceVecNormalize  # noqa: F401
