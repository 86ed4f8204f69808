"""A part evaluated as K clusters of its equations, one launch each (yask_amd/csrc/ykh_subpart.hpp): the reference's fsg bundles
12 velocity and 24 stress updates into two parts (src/stencils/FSGElasticStencil.cpp) that fit no marching kernel whole.

tests/test_reference_stencils_gpu.py holds every registered shape -- the `c<K>_...` ones included -- to the reference's outputs on
the golden grids.  Here: a ragged multi-tile grid with x-chunks, against the point kernel evaluating the whole bundle at once
(same expressions, same order: the clusters only drop equations, so the bound is rounding noise of fused multiply-adds, 2e-6
relative to the largest value), and the legality rule at run time: the clusters of a part never read what another cluster of the
part writes, so two steps through the cluster shapes equal two steps through the whole bundle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _make(stencil, size, opts):
    from yask_amd import yk_factory
    fac = yk_factory(stencil)
    s = fac.new_solution(fac.new_env())
    s.set_overall_domain_size_vec(list(size))
    assert s.apply_command_line_options(opts) == ""
    s.prepare_solution()
    for i, v in enumerate(s.get_vars()):
        v.set_elements_hash(1.5, 0.5, hash_id=i)
    return s


def _state(s, t):
    n = s.get_overall_domain_size_vec()
    out = {}
    for v in s.get_vars():
        dn = v.get_dim_names()
        if len(dn) >= 4 and dn[0] == "t":
            first = [t] + [0] * (len(dn) - 1)
            last = [t] + [n[0] - 1, n[1] - 1, n[2] - 1] + [v.get_last_misc_index(d) for d in dn[4:]]
            for k, d in enumerate(dn[4:]):
                first[4 + k] = v.get_first_misc_index(d)
            out[v.get_name()] = np.asarray(v.get_elements_in_slice(first, last), dtype=np.float64)
    return out


@pytest.mark.parametrize("stencil", ["fsg", "fsg2"])
def test_cluster_shapes_equal_the_whole_bundle_on_the_point_kernel(gpu, stencil):
    from yask_amd import yk_factory
    size, steps = (40, 37, 150), 2
    fac = yk_factory(stencil)
    probe = fac.new_solution(fac.new_env())
    nparts = probe.get_num_parts()
    per_part = [[n for n in probe.get_kernel_variant_names(p) if n.startswith("c") and n[1].isdigit()] for p in range(nparts)]
    assert sum(len(x) for x in per_part) >= 4, per_part
    ref_s = _make(stencil, size, "-hip_variant naive")
    ref_s.run_solution(0, steps - 1)
    ref = _state(ref_s, steps)
    ref_s.end_solution()
    assert ref and all(np.isfinite(a).all() for a in ref.values())
    # (-hip_variant names a shape for every part that has it: parts without it keep their default)
    names = sorted({n for x in per_part for n in x})
    for k, name in enumerate(names):
        s = _make(stencil, size, f"-hip_variant {name} -hip_xchunk {(17, 0)[k % 2]}")
        assert name in [s.get_kernel_variant(p) for p in range(nparts)]
        s.run_solution(0, steps - 1)
        got = _state(s, steps)
        s.end_solution()
        for vn, a in ref.items():
            err = np.abs(got[vn] - a).max() / max(1e-30, np.abs(a).max())
            assert err <= 2e-6, (stencil, name, vn, err)


@pytest.mark.parametrize("stencil,world,nr,g", [("fsg", 2, (1, 2, 1), (24, 40, 48)), ("fsg_abc", 8, (2, 2, 2), (48, 56, 64))])
def test_clusters_and_box_lists_over_ranks_equal_one_rank(gpu, stencil, world, nr, g, monkeypatch):
    """A decomposed run launches the cluster kernels on exterior slabs and interior boxes (fsg, two ranks); fsg_abc over 8 ranks: every
    rank finds the full boxes of the absorbing shell inside ITS domain (a corner rank owns three faces of it) and walks them.  Same
    shapes on every rank and in the one-rank run: the assembled result is the one-rank result bit for bit."""
    import test_transport_gpu as T
    monkeypatch.setenv("YASK_TEST_TRANSPORT", "ipc")
    monkeypatch.setenv("YASK_TEST_EXTRA_OPTS", "")
    steps = 2
    parts = T._run_ranks(world, "run", stencil=stencil, g=g, nr=nr, steps=steps)
    full = T._assemble(parts, stencil, g)
    one = T._one_rank(stencil, g, steps)
    for n in T.FIELDS[stencil]:
        assert np.isfinite(one[n]).all(), n
        assert np.array_equal(full[n], one[n]), (n, float(np.abs(full[n] - one[n]).max()))
