import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """No test waits for ever: with pytest-timeout present (it is in this image) every test gets a 15-minute ceiling unless it
    sets its own -- the longest CPU test takes about a minute, the longest GPU test (the 2^23 MSM, the k = 14 proof) well under
    one; a hang then costs a failure with a stack dump, not the rest of the run."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(900))


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
