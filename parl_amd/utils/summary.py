"""parl.utils.summary stand-in: add_scalar() appends to an in-memory / CSV log (tensorboardX and
VisualDL are not available here; parl/utils/summary.py:15-18 picks whichever is installed)."""
import csv
import os


class _Summary(object):
    def __init__(self):
        self.rows = []
        self.path = os.environ.get('PARL_AMD_SUMMARY_CSV')

    def add_scalar(self, tag, scalar_value, global_step=None):
        self.rows.append((tag, float(scalar_value), global_step))
        if self.path:
            with open(self.path, 'a', newline='') as f:
                csv.writer(f).writerow([tag, float(scalar_value), global_step])


summary = _Summary()
