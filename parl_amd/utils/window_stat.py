"""WindowStat — windowed mean/min/max (parl/utils/window_stat.py:20-54)."""
import numpy as np

__all__ = ['WindowStat']


class WindowStat(object):
    def __init__(self, window_size):
        self.items = [None] * window_size
        self.idx = 0
        self.count = 0

    def add(self, obj):
        self.items[self.idx] = obj
        self.idx = (self.idx + 1) % len(self.items)
        self.count += 1

    def _valid(self):
        return self.items[:self.count] if self.count < len(self.items) else self.items

    @property
    def mean(self):
        return np.mean(self._valid()) if self.count > 0 else None

    @property
    def min(self):
        return np.min(self._valid()) if self.count > 0 else None

    @property
    def max(self):
        return np.max(self._valid()) if self.count > 0 else None
