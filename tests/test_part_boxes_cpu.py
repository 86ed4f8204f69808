"""Host logic of the full-box lists (yask_amd/csrc/ykh_boxes.hpp `decompose_full_boxes`, the algorithm behind
Solution::find_part_boxes): the region is known only through two reductions -- count + bounding box of a query box, per-index
profiles -- which on the GPU are two kernels and here are brute force over a 40 x 36 x 48 mask (tests/cpp/boxes_probe.cpp, plain g++).
The reference's counterpart scans the points (StencilPartBase::find_bounding_boxes, src/kernel/lib/setup.cpp:1235-1500) and its
contract is the one asserted here: the boxes are full, do not overlap and together hold every valid point."""
import json
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def results(tmp_path_factory):
    if not shutil.which("g++"):
        pytest.skip("no g++ here")
    exe = tmp_path_factory.mktemp("boxes") / "boxes_probe"
    subprocess.run(["g++", "-O2", "-std=c++17", f"-I{ROOT / 'yask_amd' / 'csrc'}", "-o", str(exe), str(ROOT / "tests" / "cpp" / "boxes_probe.cpp")], check=True)
    return {r["region"]: r for r in json.loads(subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout)}


@pytest.mark.parametrize("region,boxes", [("shell", 6), ("box_complement", 6), ("plane_with_margins", 1), ("l_shape", 2), ("two_boxes", 2), ("solid", 1)])
def test_regions_made_of_slabs_become_exact_box_lists(results, region, boxes):
    r = results[region]
    assert r["found"] == 1 and r["boxes"] == boxes, r
    assert r["full"] == 1 and r["disjoint"] == 1 and r["covering"] == 1, r
    # the cost: a shell resolves in three levels -- 3 profiles and 10 queries, whatever the size of the grid
    if region in ("shell", "box_complement"):
        assert r["profiles"] == 3 and r["queries"] <= 10, r


@pytest.mark.parametrize("region", ["slanted", "checkerboard", "staircase"])
def test_other_regions_are_left_to_the_per_point_predicate(results, region):
    r = results[region]
    assert r["found"] == 0, r               # (the caller then keeps the point kernel with the condition evaluated per point)
    assert r["profiles"] <= 2, r            # ... and finds out quickly


@pytest.mark.parametrize("region,boxes", [("ring2d", 4), ("ring2d_open_side", 3)])
def test_two_d_rings_become_strips(results, region, boxes):
    """round 6 (`ring_strips`): the complement of an interior condition in a 2-D grid -- swe2d defines 31 of its scratch vars on such a
    ring -- is a box minus a solid hole: two reductions (the points where the condition holds, the points of their box where it does
    not) and at most four strips"""
    r = results[region]
    assert r["found"] == 1 and r["boxes"] == boxes and r["full"] == 1 and r["disjoint"] == 1 and r["covering"] == 1, r


@pytest.mark.parametrize("region", ["two_holes2d", "l_hole2d", "solid2d"])
def test_two_d_regions_that_are_not_rings_are_refused(results, region):
    assert results[region]["found"] == 0 and results[region]["boxes"] == 0, results[region]
