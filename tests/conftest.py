import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def synth():
    from cartographer_amd import synth as s
    s.lib()
    return s


@pytest.fixture
def debug():
    """Test switches of the library (include/cartographer_mi355x_debug.h): `debug(name=value, ...)`
    selects a device path / verification mode for this test; everything is reset afterwards."""
    from cartographer_amd import _lib
    yield _lib.debug_set
    _lib.debug_reset()
