"""parl_amd.env — on-device counterparts of parl.env (vector_env.py, atari_wrappers.py)."""
from .device_vector_env import DeviceVectorEnv, find_rom, GAMES  # noqa: F401
from .vec_normalize import DeviceVecNormalize  # noqa: F401
from . import atari_wrappers, vector_env  # noqa: F401,E402
from .vector_env import VectorEnv  # noqa: F401,E402
from .atari_wrappers import wrap_deepmind, MonitorEnv, get_wrapper_by_cls  # noqa: F401,E402
