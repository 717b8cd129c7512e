"""paddle.io.DataLoader.from_generator as examples/IMPALA/train.py:129-133 uses it: a bounded prefetch queue in
front of a batch generator.  Here the generator is simply iterated (its batches are numpy arrays that
agent.learn uploads itself, atari_agent.py:58-63); the learner thread blocks in the generator's own queue."""


class _GeneratorLoader(object):
    def __init__(self, capacity):
        self.capacity = capacity
        self._reader = None

    def set_batch_generator(self, reader, places=None):
        self._reader = reader
        return self

    def __call__(self):
        return iter(self._reader())

    __iter__ = __call__


class DataLoader(object):
    @staticmethod
    def from_generator(feed_list=None, capacity=None, use_double_buffer=True, iterable=True, return_list=False,
                       use_multiprocess=False, drop_last=True):
        return _GeneratorLoader(capacity)
