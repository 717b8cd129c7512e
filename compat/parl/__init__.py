"""`import parl` for scripts written against PaddlePaddle/PARL, served by parl_amd.

Put this directory (compat/) on PYTHONPATH and the reference's own example scripts run unchanged
on the MI355X path (tests/test_reference_scripts.py executes
/root/reference/benchmark/torch/a2c/{train,actor,atari_agent,atari_model}.py by path):

    parl.Model / Algorithm / Agent, parl.algorithms.{A2C, IMPALA, PPO}, parl.remote_class / connect,
    parl.env.atari_wrappers.{wrap_deepmind, MonitorEnv, get_wrapper_by_cls},
    parl.env.vector_env.VectorEnv, parl.utils.{logger, summary, rl_utils, scheduler, window_stat,
    time_stat}

`parl` and every `parl.x.y` are ALIASES of the parl_amd modules (same module objects, no second
copy): an import hook maps the names."""
import importlib
import importlib.abc
import importlib.util
import sys

import parl_amd


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path=None, target=None):
        if name != 'parl' and not name.startswith('parl.'):
            return None
        real = 'parl_amd' + name[4:]
        try:
            importlib.import_module(real)
        except ImportError:
            return None
        return importlib.util.spec_from_loader(name, self)

    def create_module(self, spec):
        return sys.modules['parl_amd' + spec.name[4:]]

    def exec_module(self, module):
        pass


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
sys.modules['parl'] = parl_amd
