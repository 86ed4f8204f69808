"""The harness (yask_amd/harness.py, counterpart of src/kernel/yask_main.cpp) prints the log keys the
reference's tooling greps (src/kernel/yask.sh:595-613, utils/lib/YaskUtils.pm:36-110) and its -validate mode
(tuned kernel vs the generic point kernel, the role of the reference's scalar run_ref) passes."""
import re
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _run(args):
    return subprocess.run([sys.executable, "-m", "yask_amd.harness"] + args, cwd=ROOT, capture_output=True, text=True, timeout=600)


@pytest.mark.parametrize("stencil,size", [("iso3dfd", "96"), ("ssg", "64"), ("3axis", "72")])
def test_harness_validate_and_log_keys(gpu, stencil, size):
    r = _run(["-stencil", stencil, "-g", size, "-trial_steps", "4", "-num_trials", "3", "-validate"])
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "TEST PASSED on rank 0." in r.stderr
    assert "YASK DONE." in r.stdout
    for key in ("best-throughput (num-points/sec):", "mid-throughput (num-points/sec):", "best-elapsed-time (sec):",
                "best-num-steps-done:", "num-trials:", "ave-throughput (num-points/sec):"):
        assert key in r.stdout, key
    m = re.search(r"best-num-steps-done:\s+(\d+)", r.stdout)
    assert m and int(m.group(1)) == 4


def test_harness_rejects_unknown_options(gpu):
    r = _run(["-stencil", "iso3dfd", "-g", "64", "-no_such_option", "3"])
    assert r.returncode != 0
    assert "extraneous parameter(s)" in r.stderr and "YASK Kernel: YASK error" in r.stderr
