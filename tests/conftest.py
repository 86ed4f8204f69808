import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
