import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) when no device is visible, so the
    # CPU-only run `-m "not gpu"` and an accidental full run both stay green.
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_sessionfinish(session, exitstatus):
    """The fp32 comparisons record what they measured (test_gpu_parity.STATS);
    with PDDP_DUMP_STATS=<path> the rows are written out as JSON - the source
    of profiles/r02_fp32_parity_rows.json."""
    path = os.environ.get("PDDP_DUMP_STATS")
    mod = sys.modules.get("test_gpu_parity")
    if path and mod is not None and getattr(mod, "STATS", None):
        import json
        with open(path, "w") as fh:
            json.dump(mod.STATS, fh)


@pytest.fixture(autouse=True)
def _sync_after_each_gpu_test(request):
    """PDDP_SYNC_EACH=1: a device synchronisation after every test, so that an
    asynchronous GPU fault is reported in the test that caused it (a debugging
    aid; off by default)."""
    yield
    if os.environ.get("PDDP_SYNC_EACH"):
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            print("[synced after %s]" % request.node.name, flush=True)
