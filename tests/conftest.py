import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / the round-end driver)")


@pytest.fixture(scope="session")
def oracle_lib():
    """The CPU oracle (test infrastructure). Built on demand with gcc."""
    from oracle import oracle as O
    O.load()
    return O


@pytest.fixture(scope="session")
def engine():
    """The HIP engine package; GPU tests fail loudly if libmplx.so is absent."""
    import motion_primitive_library_amd as m
    m._abi.lib()
    return m
