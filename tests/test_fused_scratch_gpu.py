"""Scratch vars on chip (csrc/ykh_fused.hpp; SURVEY.md section 8 row f1: "scratch vars as LDS/registers").

The reference computes scratch vars per micro-block into per-thread arrays that stay in cache, then the equations that read them
(src/kernel/lib/stencil_calc.cpp:40-289).  Here a 2-D solution's scratch stages and the stage they feed run as ONE kernel per step:
a workgroup evaluates, level by level, every scratch part over its tile grown by the scratch halos into LDS slots (shared by vars
whose live ranges do not overlap), then the consuming parts over the tile.  wave2d (src/stencils/Wave2dStencil.cpp: 12 scratch parts
+ 3) and swe2d (SWE2dStencil.cpp: 61 + 4, 39 scratch vars in 12 slots) are the reference's `2d-tests2` (src/kernel/Makefile:1126-1128).

Checker: outputs of the UNMODIFIED reference (tests/golden/, one-tile grids and the ragged 40 x 520 multi-tile ones), fp32 bound
max|gpu - ref| / max|ref| <= 2e-5 as for every generic solution; the fused path is forced (YASK_HIP_FUSE_SCRATCH=1) and asserted to be
the one that ran, and compared with the part-by-part path (=0) as a tight second bound."""
import json
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden"
INDEX = json.load(open(G / "index.json"))
CASES = sorted(n for n in INDEX if INDEX[n].get("generic") and INDEX[n]["stencil"] in ("wave2d", "swe2d", "test_scratch_2d"))


def _slice(soln, var, t):
    dn = var.get_dim_names()
    dom = soln.get_domain_dim_names()
    sdim = soln.get_step_dim_name()
    first, last, squeeze = [], [], None
    for i, d in enumerate(dn):
        if d == sdim:
            first.append(t); last.append(t); squeeze = i
        elif d in dom:
            first.append(var.get_first_rank_domain_index(d)); last.append(var.get_last_rank_domain_index(d))
        else:
            first.append(var.get_first_misc_index(d)); last.append(var.get_last_misc_index(d))
    if not dn:
        return np.asarray(var.get_element([]))
    a = var.get_elements_in_slice(first, last)
    return a[0] if squeeze == 0 else a


def _run(meta, fuse, monkeypatch, one_call=True):
    from yask_amd import yk_factory
    monkeypatch.setenv("YASK_HIP_FUSE_SCRATCH", str(fuse))
    fac = yk_factory(meta["stencil"])
    s = fac.new_solution(fac.new_env())
    s.set_overall_domain_size_vec(meta["size"])
    s.prepare_solution()
    for i, v in enumerate(s.get_vars()):
        off, sc = meta.get("init_vars", {}).get(v.get_name(), meta["init"])
        v.set_elements_hash(off, sc, hash_id=i)
    if one_call:
        s.run_solution(0, meta["steps"] - 1)
    else:
        for t in range(meta["steps"]):
            s.run_solution(t)
    return s


@pytest.mark.parametrize("name", CASES)
def test_fused_scratch_groups_match_the_reference(gpu, name, monkeypatch):
    meta = INDEX[name]
    z = np.load(G / f"{name}.npz")
    fused = _run(meta, 1, monkeypatch)
    groups = fused.get_fused_groups()
    assert len(groups) == 1 and groups[0]["lds_bytes"] <= 160 * 1024, groups
    if meta["stencil"] == "swe2d":
        assert groups[0]["parts"] == 65 and groups[0]["scratch_vars"] == 39 and groups[0]["lds_slots"] <= 16 and groups[0]["tile"] in ((32, 64), (32, 32), (16, 64), (8, 64)), groups
    if meta["stencil"] == "wave2d":
        assert groups[0]["parts"] == 15 and groups[0]["scratch_vars"] == 6, groups
    plain = _run(meta, 0, monkeypatch)
    assert plain.get_fused_groups() == []
    stepwise = _run(meta, 1, monkeypatch, one_call=False)          # one run_solution() call per step: no step graph, same kernel
    checked = 0
    for key in meta["arrays"]:
        vname, t = key.split("@")
        ref = z[key].astype(np.float64)
        assert np.isfinite(ref).all()
        scale = max(1e-30, np.abs(ref).max())
        got = np.asarray(_slice(fused, fused.get_var(vname), int(t)), dtype=np.float64)
        assert got.shape == ref.shape, (key, got.shape, ref.shape)
        assert np.abs(got - ref).max() / scale <= 2e-5, (key, "fused vs reference", np.abs(got - ref).max() / scale)
        pl = np.asarray(_slice(plain, plain.get_var(vname), int(t)), dtype=np.float64)
        assert np.abs(got - pl).max() / scale <= 2e-6, (key, "fused vs part by part")
        sw = np.asarray(_slice(stepwise, stepwise.get_var(vname), int(t)), dtype=np.float64)
        assert np.array_equal(got, sw), (key, "one call vs one call per step")
        checked += 1
    assert checked == len(meta["arrays"])
    for s in (fused, plain, stepwise):
        s.end_solution()


def test_prepare_solution_decides_by_timing_and_reports_it(gpu, monkeypatch):
    """without the environment variable prepare_solution() times a step both ways and keeps the faster (measured, profiles/r6_fused:
    the fused kernel is instruction-bound -- it wins 1.6x for wave2d at 8-byte reals and 1.2x for swe2d, and is level with 15 sweeps
    for fp32 wave2d): whichever it keeps, the solution says so, and solutions the 2-D kernel does not take never use it"""
    from yask_amd import yk_factory
    monkeypatch.delenv("YASK_HIP_FUSE_SCRATCH", raising=False)
    for stencil in ("wave2d", "swe2d", "wave2d_f64"):
        fac = yk_factory(stencil)
        s = fac.new_solution(fac.new_env())
        s.set_overall_domain_size_vec([2048, 2048])
        s.prepare_solution()
        g = s.get_fused_groups()
        assert len(g) in (0, 1), (stencil, g)
        if stencil == "wave2d_f64":
            assert len(g) == 1, "8-byte reals: twice the scratch traffic per sweep, the fused kernel wins by 1.6x"
        s.end_solution()
    # three domain dims: never fused (test_scratch_3d has a fuse group in its header, the 2-D kernel does not take it)
    fac = yk_factory("test_scratch_3d")
    s = fac.new_solution(fac.new_env())
    s.set_overall_domain_size_vec([64, 64, 64])
    monkeypatch.setenv("YASK_HIP_FUSE_SCRATCH", "1")
    s.prepare_solution()
    assert s.get_fused_groups() == []
    s.end_solution()


@pytest.mark.parametrize("size", [(1, 1), (3, 5), (17, 130), (33, 65), (64, 64), (65, 129)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("stencil", ["wave2d", "swe2d"])
def test_fused_equals_part_by_part_on_awkward_sizes(gpu, stencil, size, monkeypatch):
    """grids smaller than a tile, one point, one row more than a tile, ragged in both dims: the fused kernel against the part-by-part path
    (which the reference goldens pin at 40 x 36 and 40 x 520).  swe2d's boundary rings are wider than the smallest of these grids: the
    ring / strip logic must cope with holes that do not exist."""
    meta = dict(INDEX[[n for n in CASES if INDEX[n]["stencil"] == stencil][0]], size=list(size), steps=3)
    fused = _run(meta, 1, monkeypatch)
    assert len(fused.get_fused_groups()) == 1
    plain = _run(meta, 0, monkeypatch)
    for v in fused.get_vars():
        dn = v.get_dim_names()
        if len(dn) != 3 or dn[0] != fused.get_step_dim_name():
            continue
        t = v.get_last_valid_step_index()
        a = np.asarray(_slice(fused, v, t), dtype=np.float64)
        b = np.asarray(_slice(plain, plain.get_var(v.get_name()), t), dtype=np.float64)
        assert np.isfinite(b).all(), v.get_name()
        assert np.abs(a - b).max() <= 2e-6 * max(1e-30, np.abs(b).max()), (v.get_name(), size)
    fused.end_solution()
    plain.end_solution()
