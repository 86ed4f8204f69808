"""Regenerates the golden fixtures in this directory from the UNMODIFIED reference.

Runs only in the dev container: needs oracle/_ref (built by oracle/Makefile from /root/reference).
Each fixture is the output of oracle/_ref/bin/ref_driver.<tag>.<arch>.exe, i.e. of the reference's
own optimized CPU kernel driven through its public yk_* API, on the logical-index hash inputs that
oracle/oracle.py and the HIP runtime reproduce bit-for-bit.

    python tests/golden/make_golden.py
"""
import json
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import oracle as O  # noqa: E402

REF = ROOT / "oracle" / "_ref" / "bin"
HERE = Path(__file__).resolve().parent

CASES = [
    # name, driver tag, stencil key, size, steps, init overrides
    ("iso3dfd_32x24x40_s3", "iso3dfd", "iso3dfd", (32, 24, 40), 3),
    ("iso3dfd_20x52x36_s5", "iso3dfd", "iso3dfd", (20, 52, 36), 5),
    ("3axis_fp64_24x28x32_s4", "3axis_fp64", "3axis", (24, 28, 32), 4),
    ("ssg_24x20x28_s3", "ssg", "ssg", (24, 20, 28), 3),
]


def main():
    arch = "avx512" if "avx512f" in open("/proc/cpuinfo").read() else "avx2"
    index = {}
    for name, tag, key, size, steps in CASES:
        exe = REF / f"ref_driver.{tag}.{arch}.exe"
        if not exe.exists():
            print("skip (not built):", exe)
            continue
        with tempfile.TemporaryDirectory() as td:
            cmd = [str(exe), "-g", *map(str, size), "-steps", str(steps), "-out", f"{td}/o"]
            for v, (off, sc) in O.DEFAULT_INIT[key].items():
                cmd += ["-init", f"{v}:{off}:{sc}"]
            subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            dump = O.load_ref_dump(f"{td}/o")
        arrays = {f"{n}@{t}": a for (n, t), a in dump.items()}
        np.savez(HERE / f"{name}.npz", **arrays)
        index[name] = {"stencil": key, "size": list(size), "steps": steps, "arch": arch,
                       "arrays": sorted(arrays), "init": O.DEFAULT_INIT[key]}
        print("wrote", name, {k: v.shape for k, v in arrays.items()})
    json.dump(index, open(HERE / "index.json", "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
