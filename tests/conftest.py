import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    return load


@pytest.fixture
def dgx_dev():
    """dgx_dev_set(key, value) for one test (forced GEMM tile / split-K / kernel form: include/divergen_hip.h); everything back to the
    library's own plan afterwards."""
    from divergen_amd import _lib
    L = _lib.lib()

    def setk(key, value):
        assert L.dgx_dev_set(key.encode(), int(value)) == 0, key
    yield setk
    L.dgx_dev_set(b"reset", 0)
