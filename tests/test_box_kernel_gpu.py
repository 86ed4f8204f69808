"""The plane-ring marching kernel (yask_amd/csrc/ykh_box.hpp) -- parts that read a var at many mixed offsets: the reference's
`cube` (5x5x5 box), `3plane` (three 7x7 planes), `3axis_with_diags` and `tti` (src/stencils/SimpleStencils.cpp, TTIStencil.cpp).

tests/test_reference_stencils_gpu.py already holds EVERY registered shape of every part to the reference's own outputs, on the
golden grids (20 x 18 x 24: one tile).  Here the things a one-tile grid cannot show: several tiles and x-chunks with ragged edges
(ring slots wrap, tile halos come from neighbouring tiles' points, the prefetched plane runs past the chunk), groups whose ring
does not fit the LDS (tti: global loads where used), and a decomposed run (exterior slabs / interior boxes at odd offsets, halos
from neighbours incl. edges and corners).  The reference for each is the always-legal point kernel (`naive`: generated
calc_scalar-style code, csrc/ykh_device.hpp), itself held to the reference's outputs by the golden test; the expression is
evaluated in the same order by both, so the bound is tight: 2e-6 relative to the largest value (fp32)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
STENCILS = ["cube", "3plane", "3axis_with_diags", "tti"]


def _make(stencil, size, opts):
    from yask_amd import yk_factory
    fac = yk_factory(stencil)
    s = fac.new_solution(fac.new_env())
    s.set_overall_domain_size_vec(list(size))
    assert s.apply_command_line_options(opts) == ""
    s.prepare_solution()
    for i, v in enumerate(s.get_vars()):
        v.set_elements_hash(1.5, 0.5, hash_id=i)          # (as tests/golden/make_golden.py initialises the reference)
    return s


def _written(s, t):
    out = {}
    n = s.get_overall_domain_size_vec()
    for v in s.get_vars():
        if v.get_num_dims() == 4:
            out[v.get_name()] = np.asarray(v.get_elements_in_slice([t, 0, 0, 0], [t, n[0] - 1, n[1] - 1, n[2] - 1])[0], dtype=np.float64)
    return out


def _box_names(stencil):
    from yask_amd import yk_factory
    fac = yk_factory(stencil)
    s = fac.new_solution(fac.new_env())
    return [n for n in s.get_kernel_variant_names(0) if n.startswith("box_")]


@pytest.mark.parametrize("stencil", STENCILS)
def test_every_plane_ring_shape_equals_the_point_kernel_on_a_ragged_multi_tile_grid(gpu, stencil):
    size, steps = (75, 45, 300), 2          # 3 x 3 tiles of 128 x 16 with ragged ends; x-chunks of 37 and 23 planes
    names = _box_names(stencil)
    assert names, "no plane-ring shape registered"
    ref_s = _make(stencil, size, "-hip_variant naive")
    ref_s.run_solution(0, steps - 1)
    ref = _written(ref_s, steps)
    ref_s.end_solution()
    assert ref and all(np.isfinite(a).all() for a in ref.values())
    for k, name in enumerate(names):
        s = _make(stencil, size, f"-hip_variant {name} -hip_xchunk {(37, 23)[k % 2]}")
        assert s.get_kernel_variant(0) == name
        s.run_solution(0, steps - 1)
        got = _written(s, steps)
        s.end_solution()
        for vn, a in ref.items():
            err = np.abs(got[vn] - a).max() / np.abs(a).max()
            assert err <= 2e-6, (stencil, name, vn, err, np.argwhere(np.abs(got[vn] - a) > 2e-6 * np.abs(a).max())[:4].tolist())


def test_the_timed_choice_is_a_plane_ring_shape_where_the_box_is_dense(gpu):
    """prepare_solution() times the registered shapes of a generic-registry part (csrc/ykh_tune.cpp); at 256^3 the plane-ring
    shapes are 2-3x faster than the point kernels on the box / plane stencils (profiles/r5_box)."""
    from yask_amd import yk_factory
    for stencil in ["cube", "3plane", "3axis_with_diags"]:
        fac = yk_factory(stencil)
        s = fac.new_solution(fac.new_env())
        s.set_overall_domain_size_vec([256, 256, 256])
        s.prepare_solution()
        assert s.get_kernel_variant(0).startswith("box_"), (stencil, s.get_kernel_variant(0))
        s.end_solution()


@pytest.mark.parametrize("world,nr", [(2, (1, 2, 1)), (8, (2, 2, 2))])
def test_cube_on_the_plane_ring_kernel_over_ranks_equals_one_rank(gpu, world, nr, monkeypatch):
    """cube reads all 124 neighbours within distance 2: a rank needs its neighbours' faces, edges AND corners, and the kernel runs
    on exterior slabs and interior boxes whose origins are no multiple of a tile.  Every rank and the one-rank run use the same
    shape, so the assembled result is the one-rank result bit for bit."""
    import test_transport_gpu as T
    monkeypatch.setenv("YASK_TEST_TRANSPORT", "ipc")
    assert T.KERNEL["cube"].startswith("-hip_variant box_")          # (the rank processes read it from that module)
    monkeypatch.setenv("YASK_TEST_EXTRA_OPTS", "")
    g, steps = (40, 44, 72), 3
    parts = T._run_ranks(world, "run", stencil="cube", g=g, nr=nr, steps=steps)
    full = T._assemble(parts, "cube", g)
    one = T._one_rank("cube", g, steps)
    assert np.isfinite(one["A"]).all()
    assert np.array_equal(full["A"], one["A"]), (float(np.abs(full["A"] - one["A"]).max()), np.argwhere(full["A"] != one["A"])[:4].tolist())
