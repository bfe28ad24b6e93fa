import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# in-process GPU tests run the library in the documented runtime configuration (set before the HIP runtime initialises in this process)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def manifest():
    with open(os.path.join(GOLDEN, "MANIFEST.json")) as f:
        return json.load(f)["cases"]


@pytest.fixture(scope="session")
def golden_manifest():
    return manifest()


@pytest.fixture(scope="session")
def ctx():
    """A libss4k_hip context on cuda:0; GPU tests FAIL (not skip) if the extension cannot run."""
    import torch
    import sharkshark4k_amd  # noqa: F401
    from sharkshark4k_amd import _capi
    assert torch.cuda.is_available(), "gpu-marked test started without a GPU"
    c = _capi.Context(0)
    yield c
    c.close()
