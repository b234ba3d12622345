import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def zk():
    import zkevm_circuits_amd as z
    return z


@pytest.fixture(scope="session")
def ctx(zk):
    c = zk.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def cref():
    from oracle import cref as c
    c.lib()
    return c
