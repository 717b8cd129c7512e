from .model import Model  # noqa: F401
from .algorithm import Algorithm  # noqa: F401
from .agent import Agent  # noqa: F401
