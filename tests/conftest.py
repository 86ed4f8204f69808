import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

# Files whose tests need two or more GPUs in one box.  They have never run on real hardware (every box this repo has seen has
# one GPU), so under `pytest -x` they must not stand between the driver and the parity tests that are known to be green: they
# are collected LAST (tests/test_collection_order_cpu.py holds this), and their default matrix is bounded (see that file).
MULTI_DEVICE_FILES = ("test_multi_device_gpu.py",)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "multi_device: needs two or more GPUs in one box; collected after every other test")


def is_multi_device_item(item) -> bool:
    return Path(str(item.fspath)).name in MULTI_DEVICE_FILES or item.get_closest_marker("multi_device") is not None


def pytest_collection_modifyitems(config, items):
    """Stable partition: everything that runs on one GPU (or none) first, the multi-device tests after it."""
    first = [it for it in items if not is_multi_device_item(it)]
    last = [it for it in items if is_multi_device_item(it)]
    items[:] = first + last


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    if not _have_gpu():
        pytest.fail("GPU test selected but no GPU is visible (these tests never fall back to the CPU)")
    # a box that received the sources only: compile the HIP kernel libraries in-tree first (no-op when they are there)
    from yask_amd import _capi
    _capi.ensure_built(("iso3dfd", "3axis", "3axis_r1", "ssg", "test_3d", "awp_abc", "swe2d"))
    return True
