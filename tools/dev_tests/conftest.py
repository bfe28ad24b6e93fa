"""Tests of dev-library-only kernels (measurement experiments that are not part of libss4k_hip.so).  Run by
tests/test_gpu_dev_kernels.py in a child process with SS4K_LIB pointing at libss4k_hip_dev.so and the experiment's switch set."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X")


@pytest.fixture(scope="session")
def ctx():
    import torch
    import sharkshark4k_amd  # noqa: F401
    from sharkshark4k_amd import _capi
    assert torch.cuda.is_available(), "gpu-marked test started without a GPU"
    assert os.environ.get("SS4K_LIB", "").endswith("libss4k_hip_dev.so"), "these tests need the dev library (SS4K_LIB)"
    c = _capi.Context(0)
    yield c
    c.close()
