"""parl_amd — the MI355X-native IMPALA / A2C actor-learner hot path behind PARL's API.

`import parl_amd as parl` gives the surface the reference's example scripts use:
parl.Model / parl.Algorithm / parl.Agent (torch flavour), parl.algorithms.{IMPALA, A2C},
parl.remote_class / parl.connect, parl.utils, parl.env.  All scan / sampling / env-step compute
is in libparl_hip.so (hand-written gfx950 kernels); there is no CPU fallback."""
__version__ = '0.1.0'

from .core import Model, Algorithm, Agent  # noqa: F401,E402
from . import algorithms  # noqa: F401,E402
from . import utils  # noqa: F401,E402
from . import env  # noqa: F401,E402
from .storage import RolloutStorage  # noqa: F401,E402
from .remote import remote_class, connect, RemoteError  # noqa: F401,E402
