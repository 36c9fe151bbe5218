import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    path = os.path.join(GOLDEN, name)
    if not os.path.exists(path):
        pytest.skip(f'golden fixture {name} missing')
    return np.load(path, allow_pickle=False)


@pytest.fixture(scope='session')
def golden():
    return load_golden


@pytest.fixture(autouse=True)
def _settle_gpu_between_tests(request):
    """GPU tests build engines, side streams and captured HIP graphs.  Drop them deterministically — after the device is idle —
    instead of whenever the garbage collector gets to them in the middle of a later test (a captured graph destroyed while
    another test replays its own graphs was seen to crash the HIP runtime once in a few full runs)."""
    yield
    if request.node.get_closest_marker('gpu') is None:
        return
    import gc
    import torch
    if torch.cuda.is_available():
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.synchronize()
