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


def _torch_first():
    # torch first, where it is installed: torch ships its own copy of the HIP runtime, and the copy that is loaded first is
    # the one the whole process uses (same SONAME) -- the other way round torch finds "no HIP GPUs" in tests that also use
    # torch for device memory / streams.  bench.py imports in this order too ("one HIP runtime per process").
    if os.environ.get("VS_AMD_LIB_PARTIAL") == "1":          # a host-only sanitizer build is under test: no HIP runtime in that process at all
        return
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass


@pytest.fixture(scope="session")
def vs():
    """the product's C ABI via ctypes; a missing library is an error, never a skip"""
    _torch_first()
    from video_stabilizer_amd import capi
    capi.lib()
    return capi


@pytest.fixture(scope="session")
def gpu_vs(vs):
    if vs.device_count() < 1:
        pytest.fail("gpu-marked test but libvs_amd sees no HIP device")
    return vs


# ---- the debug build with bounds-checked LDS / scratch indexing (tests/test_bounds_build_gpu.py runs a pytest of its own with
# VS_AMD_LIB pointing at variants/libvs_amd_bounds.so and VS_BOUNDS_BUILD=1): after EVERY test of that run the library's bounds
# record must be clean, so a violation is pinned to the test that committed it
@pytest.fixture(autouse=True)
def _bounds_record_stays_clean(request):
    yield
    if os.environ.get("VS_BOUNDS_BUILD") == "1" and request.node.get_closest_marker("gpu") is not None:
        _torch_first()                      # (a test that did all its GPU work in child processes has not loaded anything yet)
        from video_stabilizer_amd import capi
        n, first = capi.debug_bounds_check()
        assert n == 0, "out-of-bounds index in a kernel of the bounds build during this test: %s" % first
