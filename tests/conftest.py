import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "polar-codes_amd"), os.path.join(ROOT, "tests"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def built_library():
    """Make sure the in-tree C-ABI library exists (hipcc cross-compiles without a GPU)."""
    import pypolar_amd
    if not os.path.exists(pypolar_amd.library_path()):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "polar-codes_amd")])
    return pypolar_amd.load_library()


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(autouse=True)
def _throughput_kernels_for_small_test_batches():
    """The library routes calls of up to a few hundred rows to a low-latency sibling geometry (pcd_small_batch_rows).  The
    suite's batches are small on purpose, and most tests are about the throughput kernels, so new decoders get a threshold of 0
    here; the tests of the routing itself set pypolar_amd.SMALL_BATCH_ROWS = None."""
    import pypolar_amd
    old = pypolar_amd.SMALL_BATCH_ROWS
    pypolar_amd.SMALL_BATCH_ROWS = 0
    yield
    pypolar_amd.SMALL_BATCH_ROWS = old
