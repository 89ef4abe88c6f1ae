import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun / the driver's GPU tier)")


def pytest_collection_finish(session):
    """PyTorch ships its own HIP runtime; libmashmap_hip.so is linked against the one under /opt/rocm.  Whichever initialises first in
    a process is the one both use, and torch finds "no ROCm-capable device" when it comes second (seen on the GPU box with
    `pytest tests/test_gpu_index.py tests/test_gpu_fullsize.py`).  The tests that need torch on the GPU (full-size synthetic data) run
    in the same process as the ones that only need the library, in any order the caller picks: bring torch's device up first."""
    if not any(item.get_closest_marker("gpu") for item in session.items):
        return
    if not any("fullsize" in item.nodeid or "bench" in item.nodeid or "humanscale" in item.nodeid for item in session.items):
        return                                              # nothing selected imports torch on the GPU: spare the import
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda").cpu()
    except Exception:                                        # no torch / no GPU: the tests themselves say so
        pass


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


@pytest.fixture(scope="session")
def human(tmp_path_factory):
    """the human-scale inputs + the stock binary's background runs (tests/humanscale.py)"""
    import humanscale
    yield from humanscale.start(tmp_path_factory)
