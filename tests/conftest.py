import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (test infrastructure); built on demand with gcc."""
    from oracle import orb_oracle
    orb_oracle.build()
    return orb_oracle


@pytest.fixture(scope="session")
def numpy_ref():
    """The second, independently written restatement (NumPy)."""
    from oracle import orb_numpy
    return orb_numpy


@pytest.fixture(scope="session")
def tinyorb():
    """The product library through its Python mirror.  No fallback: missing .so -> error."""
    from tinyslam_amd import orb
    orb.load_library()
    return orb
