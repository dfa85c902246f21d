import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


def pytest_sessionstart(session):
    """On a GPU box: start paging librccl.so (570 MB) into the page cache with ONE sequential read in the background.  Its cold load
    by dlopen -- page faults in link order on a freshly started box -- took 200 s in round 2; the RCCL tests run last
    (pytest_collection_modifyitems), by which time the file is resident and the load takes seconds."""
    if not os.path.exists("/dev/kfd"):
        return
    import threading

    def warm():
        for path in ("/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"):
            try:
                with open(path, "rb", buffering=0) as f:
                    while f.read(8 << 20):
                        pass
                return
            except OSError:
                continue
    threading.Thread(target=warm, daemon=True).start()


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
        multi.sort(key=lambda it: "rccl" in it.nodeid)          # (stable: the RCCL tests last of all, see pytest_sessionstart)
        items[:] = rest + multi
    if os.path.exists("/dev/kfd"):
        return
    skip = pytest.mark.skip(reason="no AMD GPU on this machine (/dev/kfd absent); run `pytest -m gpu` on the MI355X box")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
