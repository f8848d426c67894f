import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    np.seterr(all='ignore')


@pytest.fixture(scope='session')
def golden_ops():
    return np.load(os.path.join(GOLDEN, 'operators_N16_L4.npz'))


@pytest.fixture(scope='session')
def golden_mtip16():
    return np.load(os.path.join(GOLDEN, 'mtip_N16_L4.npz'))


@pytest.fixture(scope='session')
def golden_variants():
    return np.load(os.path.join(GOLDEN, 'mtip_variants_N16_L4.npz'))


@pytest.fixture(scope='session')
def golden_cfg1():
    return np.load(os.path.join(GOLDEN, 'mtip_cfg1_N32_L8.npz'))


def rel_l2(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    n = np.linalg.norm(b.ravel())
    d = np.linalg.norm((a - b).ravel())
    return d / n if n > 0 else d


@pytest.fixture(autouse=True)
def _emulation_defaults(request, monkeypatch):
    """CPU tests run the kernels on the fiber emulation: the concurrent V_r replay (default on the GPU) makes consumer workgroups
    poll for their producer there, which costs minutes over the suite -- off unless a test is about it (it sets the variable itself)."""
    if request.node.get_closest_marker('gpu') is None and 'MTIP_JAC_CONC' not in os.environ:
        monkeypatch.setenv('MTIP_JAC_CONC', '0')
