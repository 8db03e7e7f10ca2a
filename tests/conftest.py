import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def hiplib():
    """The product library; GPU tests must exercise it (never a fallback)."""
    import torch
    from artemis_amd import capi
    L = capi.load()
    assert torch.cuda.is_available(), "GPU test on a box without a GPU"
    assert L.artemis_hip_device_count() >= 1
    return L
