"""TimeStat: `with stat: ...` measures the block; mean / min / max over the last `window_size` blocks.

Mirrors the interface of parl/utils/time_stat.py:22-52 (the examples' learners print `sample_time.mean` etc.).
The clock is time.perf_counter (monotonic: a wall-clock step during a run cannot produce a negative sample);
nested use of one instance is supported (a stack of start times) — the reference overwrites its single start
time and reports the inner block twice."""
import time
from collections import deque

__all__ = ['TimeStat']


def _stat(reduce):
    def get(self):
        return reduce(self._samples) if self._samples else None
    return property(get)


class TimeStat(object):
    def __init__(self, window_size=1):
        self._samples = deque(maxlen=int(window_size))
        self._starts = []

    def __enter__(self):
        self._starts.append(time.perf_counter())
        return self

    def __exit__(self, *exc):
        self._samples.append(time.perf_counter() - self._starts.pop())
        return False

    mean = _stat(lambda s: sum(s) / len(s))
    min = _stat(min)
    max = _stat(max)

    @property
    def count(self):
        return len(self._samples)
