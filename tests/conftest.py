import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def navlib():
    """libnavhip.so, built on demand (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as ge
    ge.build_navhip()
    from permafrost_engine_amd import navhip
    return navhip


@pytest.fixture(scope="session", autouse=True)
def _torch_sees_the_gpu_first():
    """PyTorch-ROCm brings its own HIP runtime; libnavhip.so links the system one.  Both coexist in one
    process when torch initialises first (what bench.py and tick.py do); the other way round torch
    reports "No HIP GPUs are available".  Tests that use torch only late in a session must not depend on
    which test file ran before them."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    yield
