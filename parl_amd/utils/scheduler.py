"""PiecewiseScheduler / LinearDecayScheduler with the behaviour of parl/utils/scheduler.py:20-98
(pinned by tests/golden/scheduler.npz, generated from the reference)."""

__all__ = ['PiecewiseScheduler', 'LinearDecayScheduler']


class PiecewiseScheduler(object):
    def __init__(self, scheduler_list):
        assert len(scheduler_list) > 0
        for (s0, _), (s1, _) in zip(scheduler_list[:-1], scheduler_list[1:]):
            assert s0 < s1, 'step of scheduler_list should be incremental.'
        self.scheduler_list = scheduler_list
        self.cur_index = 0
        self.cur_step = 0
        self.cur_value = scheduler_list[0][1]

    def step(self, step_num=1):
        assert isinstance(step_num, int) and step_num >= 1
        self.cur_step += step_num
        # at most ONE boundary is crossed per call, exactly like the reference (scheduler.py:52-56)
        if self.cur_index < len(self.scheduler_list) - 1:
            if self.cur_step >= self.scheduler_list[self.cur_index + 1][0]:
                self.cur_index += 1
                self.cur_value = self.scheduler_list[self.cur_index][1]
        return self.cur_value


class LinearDecayScheduler(object):
    def __init__(self, start_value, max_steps):
        assert max_steps > 0
        self.cur_step = 0
        self.max_steps = max_steps
        self.start_value = start_value

    def step(self, step_num=1):
        assert isinstance(step_num, int) and step_num >= 1
        self.cur_step = min(self.cur_step + step_num, self.max_steps)
        return self.start_value * (1.0 - ((self.cur_step * 1.0) / self.max_steps))
