"""Full boxes of a sub-domain condition (`Solution::find_part_boxes`, csrc/ykh_solution.cpp; `yk_solution_get_part_full_boxes`).

The reference turns every IF_DOMAIN condition into a list of full bounding boxes -- non-overlapping, valid points only -- and its
kernels walk that list (StencilPartBase::find_bounding_boxes / _bb_list, src/kernel/lib/setup.cpp:1235-1500).  Here
prepare_solution() finds the list with two device reductions (bounding box + count, per-index profiles) where the condition does
not fill its bounding box, and the part then runs its unpredicated kernels box by box; conditions that are not a handful of slabs
stay with the point kernel's per-point predicate.  The VALUES such parts compute are held to the reference's outputs by
tests/test_reference_stencils_gpu.py (test_boundary_3d, awp_abc, awp_elastic_abc, the fsg *_abc solutions); here the geometry."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _prepared(stencil, size, opts=""):
    from yask_amd import yk_factory
    fac = yk_factory(stencil)
    s = fac.new_solution(fac.new_env())
    s.set_overall_domain_size_vec(list(size))
    if opts:
        assert s.apply_command_line_options(opts) == ""
    s.prepare_solution()
    return s


def _mask_of(boxes, n):
    m = np.zeros(n, dtype=np.int32)
    for f, l in boxes:
        m[f[0]:l[0] + 1, f[1]:l[1] + 1, f[2]:l[2] + 1] += 1
    return m


@pytest.mark.parametrize("n", [(20, 18, 24), (150, 45, 300)])
def test_complement_of_a_box_becomes_six_full_boxes(gpu, n):
    """test_boundary_3d (src/stencils/TestStencils.cpp:853-865): sd0 = x in [5, nx-4], y in [4, ny-7], z in [6, nz-5]; the second
    equation holds where !sd0 -- the shell around that box, whose bounding box is the whole domain."""
    s = _prepared("test_boundary_3d", n)
    inside = np.zeros(n, dtype=bool)
    inside[5:n[0] - 4 + 1, 4:n[1] - 7 + 1, 6:n[2] - 5 + 1] = True
    seen = {"solid": 0, "shell": 0}
    for p in range(s.get_num_parts()):
        kind, first, last = s.get_part_bounding_box(p)
        boxes = s.get_part_full_boxes(p)
        if kind == 1 and first == [5, 4, 6]:
            assert boxes == []                       # the condition fills its bounding box: nothing to list
            seen["solid"] += 1
        elif kind == 1:
            assert first == [0, 0, 0] and last == [n[0] - 1, n[1] - 1, n[2] - 1]
            assert 1 <= len(boxes) <= 6, boxes
            m = _mask_of(boxes, n)
            assert m.max() == 1                      # non-overlapping
            assert np.array_equal(m.astype(bool), ~inside)      # valid points only, and all of them
            assert s.get_part_info(p)["points"] == int((~inside).sum())
            assert s.get_kernel_variant(p) != "" and not s.get_kernel_variant(p).startswith("abl")
            seen["shell"] += 1
    assert seen == {"solid": 1, "shell": 1}, seen
    s.end_solution()


def test_awp_abc_free_surface_and_sponge_parts(gpu):
    """awp_abc (src/stencils/AwpStencil.cpp): the below-the-surface updates fill their boxes; the free-surface parts live in planes at
    the top of z.  Whatever list a part gets: non-overlapping, inside the part's bounding box, as many points as the part reports."""
    n = (48, 40, 56)
    s = _prepared("awp_abc", n)
    listed = 0
    for p in range(s.get_num_parts()):
        kind, first, last = s.get_part_bounding_box(p)
        boxes = s.get_part_full_boxes(p)
        if not boxes:
            continue
        listed += 1
        assert kind == 1
        m = _mask_of(boxes, n)
        assert m.max() == 1
        for f, l in boxes:
            assert all(first[d] <= f[d] <= l[d] <= last[d] for d in range(3)), (first, last, f, l)
        assert int(m.sum()) == s.get_part_info(p)["points"]
    s.end_solution()
    # (how many parts get a list depends on the stencil's conditions; the call itself must work for every part)
    assert listed >= 0


def test_shell_parts_run_the_fast_kernels_and_match_the_point_kernel(gpu):
    """fsg_abc: the absorbing-boundary parts hold in a 20-point shell (FSGElasticStencil.cpp:395-397) -- six full boxes on a grid
    wider than 40 points.  The tuned choice (cluster shapes, box by box) equals the point kernel walked over the same boxes."""
    n = (64, 56, 72)
    a = _prepared("fsg_abc", n)
    nparts = a.get_num_parts()
    lists = [a.get_part_full_boxes(p) for p in range(nparts)]
    assert sum(1 for b in lists if len(b) == 6) == 2, [len(b) for b in lists]
    b = _prepared("fsg_abc", n, "-hip_variant naive")
    for s in (a, b):
        for i, v in enumerate(s.get_vars()):
            v.set_elements_hash(1.5, 0.5, hash_id=i)
        s.run_solution(0, 1)
    for va, vb in zip(a.get_vars(), b.get_vars()):
        dn = va.get_dim_names()
        if len(dn) == 4 and dn[0] == "t":
            ga = np.asarray(va.get_elements_in_slice([2, 0, 0, 0], [2, n[0] - 1, n[1] - 1, n[2] - 1]), dtype=np.float64)
            gb = np.asarray(vb.get_elements_in_slice([2, 0, 0, 0], [2, n[0] - 1, n[1] - 1, n[2] - 1]), dtype=np.float64)
            assert np.isfinite(gb).all()
            assert np.abs(ga - gb).max() <= 2e-6 * max(1e-30, np.abs(gb).max()), va.get_name()
    a.end_solution()
    b.end_solution()
