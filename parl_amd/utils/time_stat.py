"""TimeStat — context-manager timer with a windowed mean (parl/utils/time_stat.py:22-52)."""
import time

from .window_stat import WindowStat

__all__ = ['TimeStat']


class TimeStat(object):
    def __init__(self, window_size=1):
        self.time_samples = WindowStat(window_size)
        self._start_time = None

    def __enter__(self):
        self._start_time = time.time()

    def __exit__(self, exc_type, exc_value, tb):
        self.time_samples.add(time.time() - self._start_time)

    @property
    def mean(self):
        return self.time_samples.mean

    @property
    def min(self):
        return self.time_samples.min

    @property
    def max(self):
        return self.time_samples.max
