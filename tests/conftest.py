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
def golden_flow():
    return np.load(os.path.join(GOLDEN, 'average_flow.npz'))


@pytest.fixture(scope='session')
def golden_mtip2d():
    return np.load(os.path.join(GOLDEN, 'mtip2d_N12_M6.npz'))


@pytest.fixture(scope='session')
def golden_polar2d_rules():
    return np.load(os.path.join(GOLDEN, 'polar2d_rules.npz'))


@pytest.fixture(scope='session')
def golden_mtip2d_variants():
    return np.load(os.path.join(GOLDEN, 'mtip2d_variants_N12_M6.npz'))


@pytest.fixture(scope='session')
def golden_metrics():
    return np.load(os.path.join(GOLDEN, 'metrics_ops.npz'))


@pytest.fixture(scope='session')
def golden_polar2d():
    return np.load(os.path.join(GOLDEN, 'polar2d_ops.npz'))


@pytest.fixture(scope='session')
def golden_radial():
    return np.load(os.path.join(GOLDEN, 'radial_rules.npz'))


@pytest.fixture(scope='session')
def golden_cfg1():
    return np.load(os.path.join(GOLDEN, 'mtip_cfg1_N32_L8.npz'))


def rel_l2(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    n = np.linalg.norm(b.ravel())
    d = np.linalg.norm((a - b).ravel())
    return d / n if n > 0 else d
