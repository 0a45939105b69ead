import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _oracle_built():
    """Build the CPU oracle once per session (gcc; a few seconds)."""
    import oracle_backend
    oracle_backend.build_oracle()


@pytest.fixture(scope="session")
def gb():
    import gb25_amd
    return gb25_amd


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
