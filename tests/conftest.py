import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# The multi-process tests rendezvous on 127.0.0.1.  gloo otherwise starts by resolving the machine's hostname -- the container's may not
# resolve -- and on some boxes of the pool the first few such tests of a run took 55-93 s instead of 5 (two of four full-suite runs of
# round 6: +300 s); with the interface named no lookup is made, and a resolver that does get asked gives up after one second.
os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
os.environ.setdefault("RES_OPTIONS", "timeout:1 attempts:1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests must never be silently skipped on a GPU box; without a GPU they are skipped."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _release_gpu_state(request):
    """After every GPU test: drop what the test left behind -- captured hipGraphs and their memory pools live until their Python owners
    are collected, and a few hundred of them (the suite captures ~600 graphs) end in a segmentation fault inside the NEXT capture
    (hipGraph instantiation in capture_end; seen with ROCm 7.0 once the suite grew past ~510 tests)."""
    yield
    if "gpu" not in request.keywords:
        return
    import gc
    gc.collect()
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
    except Exception:
        pass
