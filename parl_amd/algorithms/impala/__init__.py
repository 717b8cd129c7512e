from . import vtrace  # noqa: F401
from .impala import IMPALA, VTraceLoss  # noqa: F401
