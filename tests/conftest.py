import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`gpu` tests need a HIP device AND the built library: on a box without one they are skipped, not failed (plain `pytest tests/`)."""
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="needs a HIP device (run on the GPU box with -m gpu)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def flame_model():
    from vhap_amd.synthetic import make_flame_model
    return make_flame_model(seed=0)


@pytest.fixture(autouse=True)
def _release_gpu_objects_between_tests(request):
    """Captured steps (hipGraphs + their private memory pools + streams) sit in reference cycles with their trackers; left to the cycle
    collector's own schedule they pile up across the GPU tests of one process.  Collect them after every GPU test."""
    yield
    if "gpu" in request.keywords and os.environ.get("VHAP_TEST_GC", "1") != "0":
        import gc
        import torch
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.empty_cache()
