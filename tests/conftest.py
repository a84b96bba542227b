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


def pytest_sessionstart(session):
    """The library is built in-tree (git-ignored): make sure it exists and is current before any
    test loads it.  The product itself never builds implicitly (shapy_amd/_lib.py)."""
    try:
        from shapy_amd import build as hip_build
        hip_build.build(force=False)
    except Exception as e:                                   # no hipcc: the C-ABI tests will say so
        print(f'[conftest] libshapy_hip.so not built: {e}')


@pytest.fixture(scope='session')
def golden_dir():
    return osp.join(ROOT, 'tests', 'golden')


@pytest.fixture(scope='session')
def synth_smplx():
    from shapy_amd.utils import synthetic as syn
    return syn.make_synthetic_smplx(0)
