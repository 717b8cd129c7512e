"""parl_amd.algorithms — torch-hosted algorithms whose scan / sampling arithmetic runs in the
gfx950 kernels (mirrors parl/algorithms/{paddle,torch})."""
from .impala.impala import IMPALA, VTraceLoss  # noqa: F401
from .impala import vtrace  # noqa: F401
from .a2c import A2C  # noqa: F401
from .ppo import PPO  # noqa: F401
