import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle_py
    oracle_py.lib()
    oracle_py.set_threads(min(8, len(os.sched_getaffinity(0))))
    return oracle_py


@pytest.fixture(scope="session", autouse=True)
def _torch_hip_first():
    """torch bundles its own HIP runtime: it must initialise BEFORE libdynogfx.so (linked against /opt/rocm) is loaded,
    otherwise torch finds no GPU afterwards (the multi-rank test sums device buffers with torch, as bench.py does)."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda")
    except Exception:
        pass
    yield
