import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    """The HIP library on a real device; fails (never skips) when the extension or GPU is missing."""
    import torch

    from cdftools_amd import hip

    assert torch.cuda.is_available(), "gpu-marked test run without a GPU"
    assert hip.device_count() > 0, "libcdftools_hip sees no HIP device"
    return hip
