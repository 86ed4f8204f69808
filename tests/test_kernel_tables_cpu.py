"""Compile-time tables of the generic kernel families, evaluated on the host (no GPU): tests/cpp/kernel_tables.hip is compiled per
solution exactly like csrc/stencil_generic.hip (host pass only, ~1.5 s each) and prints what the registry decides from.

* the plane-ring kernel (csrc/ykh_box.hpp `BoxCfg`): ring sizes, the LDS budget (groups take rings in the order of their read counts,
  what does not fit is read from global memory where used), which parts qualify (more than MAX_MIXED mixed-offset reads);
* parts as clusters of equations (csrc/ykh_subpart.hpp): the reads of a cluster from the compiler target's `read_wmask`, and the
  legality rule -- no cluster reads an array another cluster of the part writes, whatever the step offset (step slots alias)."""
import json
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "yask_amd" / "csrc"
pytestmark = pytest.mark.skipif(not shutil.which("hipcc"), reason="no hipcc here")


def fused_tables(tmp_path, soln):
    tables(tmp_path, soln)
    out = subprocess.run([str(tmp_path / f"kt_{soln}"), "fused"], capture_output=True, text=True, check=True).stdout
    return json.loads(out)


def tables(tmp_path, soln):
    exe = tmp_path / f"kt_{soln}"
    r = subprocess.run(["hipcc", "-O1", "-std=c++17", "--cuda-host-only", f"-I{CSRC}", f'-DYKH_GEN_HEADER="gen/{soln}_cdna4_hip.hpp"',
                        f"-DYKH_GEN_NS=ykh_gen_{soln}", "-o", str(exe), str(ROOT / "tests" / "cpp" / "kernel_tables.hip")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    return {p["part"]: p for p in json.loads(out)}


def test_plane_ring_tables(tmp_path):
    cube = tables(tmp_path, "cube")["part_1"]
    assert (cube["reads"], cube["mixed"], cube["box_eligible"]) == (125, 112, 1)
    # 5 x 5 x 5 around the point: planes x-2 .. x+2 + one free slot, 16 + 4 rows of 128 + 2 x 4 floats
    assert cube["box_128x16"] == {"lds": 6 * 20 * 136 * 4, "ring_reads": 125, "xover": 5}
    assert cube["box_128x8_80k"]["lds"] == 6 * 12 * 136 * 4
    tti = tables(tmp_path, "tti")["part_1"]
    assert (tti["reads"], tti["mixed"]) == (140, 68)
    # u(t) and v(t), 47 reads each, take the rings; the four 10-read coefficient arrays fit only two at a time in the whole LDS
    assert tti["box_128x16"] == {"lds": 2 * 6 * 20 * 136 * 4, "ring_reads": 94, "xover": 5}
    assert tti["box_128x8_80k"] == {"lds": 2 * 6 * 12 * 136 * 4, "ring_reads": 94}
    assert tti["box_128x8"]["ring_reads"] == 114 and tti["box_128x8"]["lds"] <= 160 * 1024
    # a star stencil has no mixed reads: the registry never gives it a plane-ring shape (count_mixed <= MAX_MIXED = 8)
    iso = tables(tmp_path, "iso3dfd")["part_1"]
    assert iso["mixed"] == 0 and iso["march_eligible"] == 1
    t3 = tables(tmp_path, "test_3d")["part_1"]
    assert t3["mixed"] <= 8


def test_equation_clusters_tables(tmp_path):
    fsg = tables(tmp_path, "fsg")
    v, s = fsg["part_1"], fsg["part_2"]
    assert (v["writes"], v["reads"], s["writes"], s["reads"]) == (12, 296, 24, 435)
    # four clusters = the four sub-grids: 3 velocity / 6 stress components each
    assert v["k4"]["legal"] == 1 and v["k4"]["c0_writes"] == 3 and v["k4"]["c0_march_ok"] == 1 and v["k4"]["c0_lds_v2_128x8"] <= 160 * 1024
    assert s["k4"]["legal"] == 1 and (s["k4"]["c0_reads"], s["k4"]["c0_writes"]) == (162, 6)
    assert s["k4"]["c0_lds_v2_128x8"] > 160 * 1024          # every stress equation needs all nine velocity derivatives: no marching kernel
    assert s["k2"]["legal"] == 1 and s["k2"]["c0_writes"] == 12
    # fsg2 keeps its components in two vars with a misc dim: storage identity is (var, misc indices), not the var alone
    fsg2 = tables(tmp_path, "fsg2")
    assert fsg2["part_2"]["k4"]["legal"] == 1 and fsg2["part_1"]["k4"]["legal"] == 1
    # tti's two equations read each other's var (u and v, at other step offsets: the step slots alias): not separable
    tti = tables(tmp_path, "tti")["part_1"]
    assert tti["k2"]["legal"] == 0


def test_partial_dim_tables_at_offsets_are_legal_on_the_plane_ring_kernel(tmp_path):
    """round 6: groups read at an offset may be vars over a SUBSET of the domain dims (kind 4: loaded through their own strides where
    used, never given a ring); written groups still have to be full.  test_partial_3d (TestStencils.cpp): A and H get rings
    (9 reads each), the 1-D / 2-D tables B .. G, K, L count as served -- 59 of its 79 reads were what kept it on the point kernels."""
    p = tables(tmp_path, "test_partial_3d")["part_1"]
    assert (p["reads"], p["box_eligible"]) == (79, 1) and p["mixed"] > 8
    assert 2 * p["box_128x16"]["ring_reads"] >= p["reads"] and p["box_128x16"]["lds"] <= 160 * 1024
    assert p["box_128x16"]["ring_reads"] >= 18 + 50


def test_two_d_parts_lift_to_one_plane_of_the_three_d_families(tmp_path):
    """ykh_lift2d.hpp: a 2-D part's reads (d0, d1, 0) become (0, d0, d1) -- one x plane, d1 along the lanes"""
    w = tables(tmp_path, "wave2d")
    for name, p in w.items():
        assert p["lift2d_shape"] == 1, name
        assert p["lifted"]["reads"] == p["reads"] and p["lifted"]["nonzero_dx"] == 0, name
    bf = tables(tmp_path, "box_filter")["part_1"]
    # the image filters read a dense 2-D neighbourhood: mixed reads in the lifted (y, z) plane -> the plane-ring kernel on ONE plane
    assert bf["lifted"]["mixed"] > 8 and bf["lifted"]["box_eligible"] == 1 and bf["lifted"]["box_128x16"]["xover"] == 1
    assert bf["lifted"]["box_128x16"]["ring_reads"] == bf["reads"]
    # a 3-D part with z offsets is not a lifted shape
    assert tables(tmp_path, "cube")["part_1"]["lift2d_shape"] == 0


def test_fused_scratch_group_plans(tmp_path):
    """csrc/ykh_fused.hpp: the compiler target lists a run of scratch stages + the stage they feed with the level of every part; the plan
    gives every scratch var an LDS slot for its live range (first write .. last read, in levels) and re-uses slots across ranges."""
    w = fused_tables(tmp_path, "wave2d")
    assert len(w) == 1 and (w[0]["parts"], w[0]["levels"], w[0]["scratch_vars"], w[0]["slots"]) == (15, 3, 6, 6) and w[0]["ok_16x64"] == 1
    assert w[0]["lds_16x64"] == 6 * (16 + 2) * (64 + 2) * 4
    s = fused_tables(tmp_path, "swe2d")
    assert len(s) == 1 and (s[0]["parts"], s[0]["levels"], s[0]["scratch_vars"]) == (65, 17, 39)
    assert s[0]["slots"] <= 16 and s[0]["halo"] == [5, 4, 5, 4]            # 39 vars, a dozen alive at a time
    assert s[0]["lds_16x64"] == s[0]["slots"] * (16 + 9) * (64 + 9) * 4 <= 160 * 1024 and s[0]["ok_16x64"] == 1
    # a 3-D solution's scratch group is listed by the compiler target but is not a shape the 2-D kernel takes (reads with a third offset)
    t3 = fused_tables(tmp_path, "test_scratch_3d")
    assert len(t3) == 1 and t3[0]["parts_ok"] == 0 and t3[0]["ok_16x64"] == 0
    assert fused_tables(tmp_path, "iso3dfd") == []
    # the invariant of the slot assignment, checked on every group the headers list: vars share a slot only when one's last use lies at
    # an earlier level than the other's first write (a barrier separates levels), and every scratch var of a group has a slot
    for soln in ("wave2d", "swe2d", "test_scratch_2d", "test_scratch_1d", "test_scratch_stages_1d", "test_scratch_boundary_1d"):
        for g in fused_tables(tmp_path, soln):
            assert g["slot_conflicts"] == 0 and g["unassigned"] == 0, (soln, g)


def test_round6_registry_rules(tmp_path):
    """Second half of round 6: which parts get plane-ring shapes (more than MAX_MIXED mixed reads, or >= 4 that are at least half of
    the reads), how many centre-only operands the late refill holds once (>= 3: `_lo` shapes are registered), and which partial-dim
    operands the compiler target's `group_dims` lets the marching kernel load once per block."""
    t3 = tables(tmp_path, "test_3d")["part_1"]
    assert (t3["reads"], t3["mixed"], t3["box_wanted"]) == (9, 8, 1)          # the 8 corners of a box: exactly MAX_MIXED
    assert tables(tmp_path, "iso3dfd")["part_1"]["box_wanted"] == 0
    assert tables(tmp_path, "cube")["part_1"]["box_wanted"] == 1
    awp = tables(tmp_path, "awp")
    v, s = awp["part_1"], awp["part_2"]
    assert v["box_wanted"] == 0 and s["box_wanted"] == 0                        # 3 / 8 mixed reads among 48 / 70
    assert v["once"] >= 3 and s["once"] >= 12                                   # vel(t) + the five partial-dim vars; stress + memory vars + coefficients
    assert v["x_invariant"] == 4 and s["x_invariant"] == 4                      # delta_t, h, cr_y, cr_z (cr_x has the marching dim)
    sp = tables(tmp_path, "iso3dfd_sponge")["part_1"]
    assert sp["x_invariant"] == 2                                               # cr_y, cr_z
    assert tables(tmp_path, "ssg")["part_1"]["x_invariant"] == 0                # every group a 3-D var

