import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun / the driver's GPU tier)")


@pytest.fixture(scope="session")
def oracle():
    import mmutil
    return mmutil.Oracle()


@pytest.fixture(scope="session")
def ref():
    import mmutil
    if not mmutil.Ref.available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    return mmutil.Ref()
