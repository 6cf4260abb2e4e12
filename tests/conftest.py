import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lsfc():
    """The product package; GPU tests call the HIP library through its C ABI."""
    import fast_solver_lippmann_schwinger_amd as pkg
    pkg.load()
    if pkg.device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests must run on the GPU box")
    return pkg


def rel_err(a, b):
    import numpy as np
    a, b = np.asarray(a).ravel(), np.asarray(b).ravel()
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))
