"""First-contact safety of the suite.

The driver runs `pytest tests -x -q -m gpu`.  tests/test_multi_device_gpu.py needs two GPUs in one box and has never run on real
hardware; collected alphabetically it would sit in FRONT of nine files of known-green parity tests, and its first failure would end
the run.  tests/conftest.py therefore moves it behind everything else; its default matrix is bounded (the full one is behind
YASK_TEST_MULTI_DEVICE_FULL=1), and an N > 1 bench line carries a cpu_baseline without re-running the reference."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _collect(extra_env=None, args=()):
    env = dict(os.environ, **(extra_env or {}))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests", "--collect-only", "-q", "-m", "gpu", *args], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return [l for l in r.stdout.splitlines() if "::" in l]


def test_multi_device_tests_are_collected_after_every_other_gpu_test():
    ids = _collect()
    files = [i.split("::")[0] for i in ids]
    multi = [k for k, f in enumerate(files) if f.endswith("test_multi_device_gpu.py")]
    assert multi, "the multi-device tests are collected (they skip themselves on a one-GPU box)"
    assert multi == list(range(len(ids) - len(multi), len(ids))), "multi-device tests must be the LAST items of a `pytest -m gpu` run"
    others = [f for f in files[:multi[0]]]
    assert others == sorted(others), "the other files keep their alphabetical order (stable partition)"
    for must_be_before in ("test_reference_stencils_gpu.py", "test_stencils_gpu.py", "test_transport_gpu.py", "test_python_api_gpu.py"):
        assert any(f.endswith(must_be_before) for f in others), must_be_before


def test_default_multi_device_matrix_is_bounded_and_the_full_one_is_opt_in():
    only = ["tests/test_multi_device_gpu.py"]
    # (the parametrisation does not depend on the devices of the box: device counts only decide skips)
    dflt = [i for i in _collect() if "test_multi_device_gpu.py" in i]
    full = [i for i in _collect({"YASK_TEST_MULTI_DEVICE_FULL": "1"}) if "test_multi_device_gpu.py" in i]
    assert len(dflt) <= 16, dflt
    assert len(full) >= 40 and set(dflt) != set(full)
    two = [i for i in dflt if "test_two_devices_equal_one_rank" in i]
    for want in ("rccl-serial", "rccl-halves", "ipc-serial", "ipc-halves", "1x1x2", "ssg"):
        assert any(want in i for i in two), (want, two)
    assert sum("test_bench_on_real_devices" in i for i in dflt) == 1
    src = (ROOT / "tests" / "test_multi_device_gpu.py").read_text()
    assert "BENCH_TIMEOUT_S = 240" in src and "timeout=BENCH_TIMEOUT_S" in src
    del only


def test_n_gt_1_bench_lines_get_a_cached_cpu_baseline(tmp_path, monkeypatch):
    sys.path.insert(0, str(ROOT))
    import bench
    monkeypatch.setattr(bench, "CPU_BASELINE_CACHE", tmp_path / "cache.json")
    cb = bench.cpu_baseline_cached()            # nothing from this box: the committed record, labelled
    assert cb and cb["cached"] and "committed record" in cb["cached_from"] and cb["kind"] in ("reference", "port") and cb["cores"] >= 1
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    bench.cpu_baseline_store({"value": 1.25, "unit": "Gpoints/s", "cores": 3, "kind": "reference", "sample": "x"})
    cb = bench.cpu_baseline_cached()
    assert cb["value"] == 1.25 and cb["cached"] and "this box's N=1 run" in cb["cached_from"]
    # another host's cache is not this box's figure
    j = json.load(open(tmp_path / "cache.json"))
    j["host"] = "some-other-box"
    json.dump(j, open(tmp_path / "cache.json", "w"))
    monkeypatch.setattr(bench, "CPU_BASELINE_RECORD", tmp_path / "none.json")
    assert bench.cpu_baseline_cached() is None
