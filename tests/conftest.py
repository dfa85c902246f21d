import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_collection_modifyitems(config, items):
    """On a machine without an AMD GPU (no /dev/kfd) the `gpu` tests are skipped instead of failing at dz_create.
    On a GPU box nothing is skipped: a missing libdreamzs.so must fail loudly there, there is no CPU fallback.

    The GPU tests that put several processes on the one device (tests/test_distributed.py: two ranks, eight ranks, RCCL bootstrap)
    run after all single-process ones: on one box in several, such a process stalled while the pytest process held the device
    (those tests have deadlines), and under `-x` a stall there must not keep the parity tests from running."""
    multi = [it for it in items if "gpu" in it.keywords and it.nodeid.startswith("tests/test_distributed.py")]
    if multi:
        rest = [it for it in items if it not in multi]
        items[:] = rest + multi
    if os.path.exists("/dev/kfd"):
        return
    skip = pytest.mark.skip(reason="no AMD GPU on this machine (/dev/kfd absent); run `pytest -m gpu` on the MI355X box")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
