"""Host-side pieces of bench.py that need no GPU: the command line parses, and the modules the N > 1 flow starts as child
processes import cleanly (a syntax error there would only show on a multi-GPU box)."""
import importlib
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def test_bench_help_lists_the_contract_flags():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1000:]
    for flag in ("--gpus", "--steps", "--warmup", "--transport", "--schedule", "--no-self-check", "--rank-grid", "--config"):
        assert flag in r.stdout, flag


def test_ipc_preflight_module_imports():
    m = importlib.import_module("yask_amd.ipc_preflight")
    assert callable(m.main)


def test_job_scripts_parse():
    for f in sorted((ROOT / "tools" / "jobs").glob("*.sh")):
        r = subprocess.run(["bash", "-n", str(f)], capture_output=True, text=True)
        assert r.returncode == 0, (f.name, r.stderr)
