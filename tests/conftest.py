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
    from tests import kmtest
    return kmtest.load_oracle()


@pytest.fixture(scope="session")
def product():
    """The product C-ABI library (kmers_amd/lib/libkmers.so), loaded through ctypes."""
    from tests import kmtest
    return kmtest.load_product()
