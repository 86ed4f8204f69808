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


# Reference test stencils (src/stencils/TestStencils.cpp, the reference's own validation matrix,
# src/kernel/Makefile:1101-1182): every var is initialised to 1.5 + 0.5*hash (positive, so that the
# sqrt/log/division stencils stay finite); var k of get_vars() uses hash id k.
GENERIC_CASES = [
    ("test_3d_18x20x22_s3", "test_3d", (18, 20, 22), 3),
    ("test_boundary_3d_20x18x24_s3", "test_boundary_3d", (20, 18, 24), 3),
    ("test_boundary_2d_40x36_s3", "test_boundary_2d", (40, 36), 3),
    ("test_scratch_3d_18x16x20_s2", "test_scratch_3d", (18, 16, 20), 2),
    ("test_scratch_boundary_1d_96_s3", "test_scratch_boundary_1d", (96,), 3),
    ("test_misc_2d_30x26_s2", "test_misc_2d", (30, 26), 2),
    ("test_stages_3d_16x18x20_s3", "test_stages_3d", (16, 18, 20), 3),
    ("test_partial_3d_16x18x20_s2", "test_partial_3d", (16, 18, 20), 2),
    ("test_stream_3d_16x18x20_s3", "test_stream_3d", (16, 18, 20), 3),
    ("test_func_1d_80_s2", "test_func_1d", (80,), 2),
    # solutions of the reference's stencil library beyond the three hot-path ones
    ("awp_abc_24x20x28_s2", "awp_abc", (24, 20, 28), 2),          # AwpStencil.cpp: 43 vars, 7 parts, boundary conditions
    ("tti_20x18x24_s2", "tti", (20, 18, 24), 2),                  # TTIStencil.cpp
    ("iso3dfd_sponge_24x20x28_s3", "iso3dfd_sponge", (24, 20, 28), 3),   # Iso3dfdStencil.cpp with sponge vars
    ("wave2d_48x40_s3", "wave2d", (48, 40), 3),                   # Wave2dStencil.cpp: 15 conditional parts
    ("ssg2_20x18x24_s2", "ssg2", (20, 18, 24), 2),                # SSGElastic2Stencil.cpp
    ("fsg2_16x14x20_s2", "fsg2", (16, 14, 20), 2),                # FSGElastic2Stencil.cpp: 81 access groups in one part
    ("cube_20x18x24_s3", "cube", (20, 18, 24), 3),                # SimpleStencils.cpp: dense 3-D cube
    # step conditions that read var values (evaluated by the kernel): even/odd steps x B(0) > B(1); t >= ti_exp()
    ("test_step_cond_1d_96_s4", "test_step_cond_1d", (96,), 4),
    # SWE2dStencil.cpp: 2-D shallow water, scratch vars; physical magnitudes so that 4 steps stay finite
    ("swe2d_40x36_s4", "swe2d", (40, 36), 4,
     {"u": (0.0, 0.1), "v": (0.0, 0.1), "e": (0.0, 0.01), "h": (1.0, 0.1), "dt": (0.002, 0.0), "dx": (0.05, 0.0), "dy": (0.05, 0.0),
      "inv_dx": (20.0, 0.0), "inv_dy": (20.0, 0.0), "g": (9.81, 0.0), "coriolis": (10.0, 0.0), "pe_offset": (0.5, 0.0),
      "ti_exp": (2.0, 0.0)}),
]
GENERIC_INIT = (1.5, 0.5)


def generic_var_names(stencil):
    import re
    txt = (ROOT / "yask_amd" / "csrc" / "gen" / f"{stencil}_cdna4_hip.hpp").read_text()
    block = txt[txt.index("static constexpr VarMeta vars[]"):txt.index("};", txt.index("static constexpr VarMeta vars[]"))]
    return [m.group(1) for m in re.finditer(r'\{"([A-Za-z_0-9]+)", \d+, .*, (true|false), (true|false)\},', block) if m.group(2) == "false"]


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
    for name, stencil, size, steps, *rest in GENERIC_CASES:
        init_vars = rest[0] if rest else {}
        exe = REF / f"ref_driver.{stencil}.{arch}.exe"
        if not exe.exists():
            print("skip (not built):", exe)
            continue
        with tempfile.TemporaryDirectory() as td:
            cmd = [str(exe), "-g", *map(str, size), "-steps", str(steps), "-out", f"{td}/o"]
            for v in generic_var_names(stencil):
                off, sc = init_vars.get(v, GENERIC_INIT)
                cmd += ["-init", f"{v}:{off}:{sc}"]
            subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            dump = O.load_ref_dump(f"{td}/o")
        arrays = {f"{n}@{t}": a for (n, t), a in dump.items()}
        np.savez_compressed(HERE / f"{name}.npz", **arrays)
        index[name] = {"stencil": stencil, "size": list(size), "steps": steps, "arch": arch, "arrays": sorted(arrays),
                       "generic": True, "init": list(GENERIC_INIT), "init_vars": {k: list(v) for k, v in init_vars.items()}}
        print("wrote", name, {k: v.shape for k, v in arrays.items()})
    json.dump(index, open(HERE / "index.json", "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
