from .scheduler import PiecewiseScheduler, LinearDecayScheduler  # noqa: F401
from .window_stat import WindowStat  # noqa: F401
from .time_stat import TimeStat  # noqa: F401
from . import rl_utils  # noqa: F401
from .rl_utils import calc_gae, calc_discount_sum_rewards  # noqa: F401
from .logger import logger  # noqa: F401
from .summary import summary  # noqa: F401
from . import machine_info  # noqa: F401
