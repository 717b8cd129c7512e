"""torch twin of examples/IMPALA/atari_model.py:21-90 (the reference file is Paddle code)."""
from parl_amd.models import AtariModel42 as AtariModel  # noqa: F401
