"""examples/PPO/storage.py of the reference: the device-resident twin lives in parl_amd.storage."""
from parl_amd.storage import RolloutStorage  # noqa: F401
