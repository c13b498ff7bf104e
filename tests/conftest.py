import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the checker (oracle) and the product library once per session.  hipcc cross-compiles without a GPU."""
    from oracle import oracle
    oracle.build()
    from linemod_pose_estimation_amd import _lib
    if not os.path.exists(_lib.SO_PATH):
        _lib.build()
    yield


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
