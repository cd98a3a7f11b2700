import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _hip_devices():
    """HIP devices visible to this process, without initialising a runtime in it (torch counts them through
    the driver's sysfs/ioctl path; device_count() does not create a context)."""
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


def pytest_collection_modifyitems(config, items):
    """Without a HIP device (this container) the `gpu` tests are skipped, not errors: plain `pytest` passes
    on a CPU box.  On a GPU box nothing is skipped, and a missing libraymarching_amd.so fails loudly."""
    if _hip_devices() > 0:
        return
    skip = pytest.mark.skip(reason="no HIP device: GPU parity tests run on the MI355X box (-m gpu)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def stock_scenes():
    from oracle import oracle as O
    return {name: O.load_sce(os.path.join(GOLDEN, "scenes", name + ".sce")) for name in ("chair", "infinity")}
