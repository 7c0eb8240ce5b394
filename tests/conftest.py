import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def pt():
    """The ctypes harness package (directory name has a hyphen)."""
    return importlib.import_module("path-tracer_amd")


@pytest.fixture(scope="session")
def orc():
    import orc as _orc
    return _orc


@pytest.fixture(scope="session")
def ctx(pt):
    """One device context for the whole GPU session (the host layer's per-process context)."""
    if not pt.gpu_available():
        pytest.fail("GPU test selected but no HIP device is visible: the product has no CPU path")
    c = pt.host().pth_context()
    assert c, pt.lib().ptx_last_error()
    return c
