import os
import sys

# tests execute reference modules from /root/reference by path (and spawn subprocesses that do): never leave a
# __pycache__ in that tree, it is read-only for this project
sys.dont_write_bytecode = True
os.environ['PYTHONDONTWRITEBYTECODE'] = '1'

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """plain `pytest tests` on a CPU-only box: everything marked gpu is skipped (not failed)"""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='needs a GPU (run with -m gpu on an MI355X box)')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: z[k] for k in z.files}


def golden_cases(d):
    """group 'case/key' entries of a golden npz into {case: {key: array}}"""
    out = {}
    for k, v in d.items():
        if '/' in k:
            c, kk = k.split('/', 1)
            out.setdefault(c, {})[kk] = v
    return out


@pytest.fixture(scope='session')
def oracle():
    """the plain-C CPU oracle (test infrastructure)"""
    from oracle import c_oracle
    c_oracle.lib()
    return c_oracle


@pytest.fixture(scope='session')
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')
