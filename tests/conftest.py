import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def golden_files(prefix):
    import glob

    return sorted(glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


@pytest.fixture(autouse=True)
def _clear_library_failure_word(request):
    """A GPU test that provokes (or dies with) a device-side failure must not leave the library's STICKY failure word set for the tests behind
    it (every later library call would fail with VAA_E_LAUNCH): poll it — the poll is what clears it — after every GPU test."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        try:
            from roboticattack_amd import _lib

            if _lib._lib is not None:
                _lib._lib.vaa_async_error()
        except Exception:
            pass
