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


@pytest.fixture
def knob():
    """Path switches of the library for the duration of one test (kmers_hip_debug_knob, include/kmers_hip.h): the library
    reads its environment once per process, so tests flip the switches through the C ABI; everything is put back (to what
    the environment says) when the test ends."""
    from kmers_amd import _lib
    lib = _lib.handle()

    def set_knob(name, value=1):
        assert lib.kmers_hip_debug_knob(name.encode(), int(value)) == 0, name

    yield set_knob
    lib.kmers_hip_debug_knob(b"reset", 0)
