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
    """the CPU restatement (test infrastructure); built on demand"""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def vs():
    """the product's C ABI via ctypes; a missing library is an error, never a skip"""
    # torch first, where it is installed: torch ships its own copy of the HIP runtime, and the copy that is loaded first is
    # the one the whole process uses (same SONAME) -- the other way round torch finds "no HIP GPUs" in tests that also use
    # torch for device memory / streams.  bench.py imports in this order too ("one HIP runtime per process").
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    from video_stabilizer_amd import capi
    capi.lib()
    return capi


@pytest.fixture(scope="session")
def gpu_vs(vs):
    if vs.device_count() < 1:
        pytest.fail("gpu-marked test but libvs_amd sees no HIP device")
    return vs
