import os
import sys
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import ommtest
    return ommtest.Lib("oracle")


@pytest.fixture(scope="session")
def product():
    """The HIP library behind the C ABI.  Fails loudly when it is missing -- there is no CPU fallback."""
    import ommtest
    return ommtest.Lib("product")
