import os
import os.path as osp
import sys

import pytest

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def golden_dir():
    return osp.join(ROOT, 'tests', 'golden')


@pytest.fixture(scope='session')
def synth_smplx():
    from shapy_amd.utils import synthetic as syn
    return syn.make_synthetic_smplx(0)
