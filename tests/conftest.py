import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def gpu_lib():
    """libraytrace_gpu.so initialised on device 0; fails loudly if the library or the GPU is missing."""
    from raytrace_amd import _lib
    L = _lib.lib()
    _lib.check(L.rt_init(0), "rt_init")
    return L
