"""torch twin of examples/A2C/atari_model.py:21-104 (identical to benchmark/torch/a2c/atari_model.py)."""
from parl_amd.models import AtariModel84 as AtariModel  # noqa: F401
