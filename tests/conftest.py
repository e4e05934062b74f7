import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The library's default tune cache (~/.cache/tsm_hip/tune_cache.txt) would carry tile choices from one test process to
# the next; tests that exercise the cache name their own file.
os.environ.setdefault('TSM_TUNE_CACHE', 'off')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(scope='session')
def sd0():
    """Seed-0 synthetic TSM-R50 state dict as torch CPU tensors."""
    from workoutdetector_amd.weights import make_state_dict, to_torch
    return to_torch(make_state_dict(0, 12))


@pytest.fixture(scope='session')
def hip_lib():
    """The C-ABI library; GPU tests must go through it, never through a fallback."""
    import torch
    from workoutdetector_amd import _lib
    from workoutdetector_amd.build import build_library
    build_library()
    assert torch.cuda.is_available(), 'gpu-marked test running without a GPU'
    return _lib.load()
