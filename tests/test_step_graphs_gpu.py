"""Captured step graphs (option -hip_step_graphs, yask_amd/csrc/ykh_solution.cpp `get_step_graph`): a one-rank
run_solution() over several steps replays ONE hipGraph that holds the launches of a whole number of step-slot periods instead
of issuing every launch from the host -- the MI355X side of the reference's step loop (`StencilContext::run_solution`,
src/kernel/lib/context.cpp:220-480), for grids whose step is launch-bound (BASELINE config 1: 128^3 x 100 steps).

A replay issues the very kernels, arguments and order of the plain loop, so the results must be BIT-identical; what is tested
is the bookkeeping around it: slot periods (iso3dfd 2 slots, ssg in place), left-over steps that do not fill a period,
continued runs starting on either slot parity, both run directions, the valid-step window, the cache (same graph re-used,
new graph after a re-prepare) and the solutions that must NOT be replayed (step conditions, step index used as a value)."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def make(stencil, size, opts, init=None):
    from yask_amd import yk_factory
    fac = yk_factory(stencil)
    soln = fac.new_solution(fac.new_env())
    soln.set_overall_domain_size_vec(list(size))
    assert soln.apply_command_line_options("-no-auto_tune " + opts) == ""
    soln.prepare_solution()
    hash_init(soln, init or stencil)
    return soln


def hash_init(soln, init):
    """`init`: a stencil of the oracle's table (same offsets / scales / hash ids as the oracle uses) or {var: (offset, scale, id)}"""
    if isinstance(init, str):
        init = {n: (*O.DEFAULT_INIT[init][n], O.VAR_IDS[init][n]) for n in O.DEFAULT_INIT[init]}
    for name, (off, sc, hid) in init.items():
        soln.get_var(name).set_elements_hash(off, sc, hash_id=hid)


def whole(soln, var, t=None):
    v = soln.get_var(var)
    n = soln.get_overall_domain_size_vec()
    dims = v.get_dim_names()
    first = [t] if dims and dims[0] == "t" else []
    return v.get_elements_in_slice(first + [0] * len(n), first + [x - 1 for x in n])[0]


@pytest.mark.parametrize("steps", [2, 7, 10, 11])
def test_iso3dfd_replayed_steps_are_bit_identical(gpu, steps):
    size = (40, 37, 70)
    g = make("iso3dfd", size, "-hip_step_graphs 1")
    p = make("iso3dfd", size, "-hip_step_graphs 0")
    g.run_solution(0, steps - 1)
    p.run_solution(0, steps - 1)
    sg, sp = g.get_stats(), p.get_stats()
    assert sg.get_num_steps_done() == steps == sp.get_num_steps_done()
    # whole periods of 2 steps in one replay; runs shorter than two periods are issued as plain launches
    assert sg.get_num_graph_steps() == (steps // 2 * 2 if steps >= 4 else 0) and sg.get_num_graph_replays() == (1 if steps >= 4 else 0)
    assert sp.get_num_graph_steps() == 0
    for t in (steps - 1, steps):
        assert np.array_equal(whole(g, "p", t), whole(p, "p", t)), t
    P = g.get_var("p")
    assert P.get_last_valid_step_index() == steps and P.get_first_valid_step_index() == steps - 1
    # ... and against the oracle, like every other kernel path
    ref = O.run_iso3dfd(size, steps)[("p", steps)]
    assert O.rel_linf(whole(g, "p", steps), ref) <= 2e-5


def test_continued_runs_on_both_slot_parities_and_the_cache(gpu):
    """Calls of different lengths, starting on even and odd steps: each (parity, length) is captured once and re-used."""
    size = (32, 30, 64)
    g = make("iso3dfd", size, "-hip_step_graphs 1")
    p = make("iso3dfd", size, "-hip_step_graphs 0")
    t = 0
    for n in (4, 5, 4, 1, 4, 6, 3):                      # starts: 0 4 9 13 14 18 24 -> both parities, repeated lengths
        g.run_solution(t, t + n - 1)
        p.run_solution(t, t + n - 1)
        t += n
        assert g.get_stats().get_num_graph_steps() == (n // 2 * 2 if n >= 4 else 0)     # (fewer than two periods: plain launches)
        for s in (t - 1, t):
            assert np.array_equal(whole(g, "p", s), whole(p, "p", s)), (t, s)
    # a re-prepare (new storage geometry) must not replay launches of the old allocation
    for s in (g, p):
        s.set_overall_domain_size_vec([36, 30, 64])
        s.prepare_solution()
        hash_init(s, "iso3dfd")
        s.run_solution(0, 5)
    assert np.array_equal(whole(g, "p", 6), whole(p, "p", 6))


def test_two_stage_in_place_solution(gpu):
    """ssg: 2 stages, nine in-place (one-slot) fields -> the period is one step, a graph holds steps x 2 launches."""
    size = (24, 20, 28)
    g = make("ssg", size, "-hip_step_graphs 1")
    p = make("ssg", size, "-hip_step_graphs 0")
    g.run_solution(0, 4)
    p.run_solution(0, 4)
    assert g.get_stats().get_num_graph_steps() == 5
    for f in ("v_bl_w", "v_tl_v", "v_tr_u", "s_bl_yz", "s_br_xz", "s_tl_xx", "s_tl_yy", "s_tl_zz", "s_tr_xy"):
        assert np.array_equal(whole(g, f, 5), whole(p, f, 5)), f


def test_fused_passes_come_first_graphs_take_nothing_from_them(gpu):
    """3axis r=1 with -hip_fuse_steps 2 runs two steps per pass; the replay only ever covers the plain loop behind it."""
    size = (40, 61, 130)
    init = {"A": (0.0, 1.0, 0)}
    g = make("3axis_r1", size, "-hip_step_graphs 1 -hip_fuse_steps 2", init)
    p = make("3axis_r1", size, "-hip_step_graphs 0 -hip_fuse_steps 2", init)
    g.run_solution(0, 8)
    p.run_solution(0, 8)
    assert g.get_stats().get_num_fused_passes() == 4
    for t in (8, 9):
        assert np.array_equal(whole(g, "A", t), whole(p, "A", t))
    g2 = make("3axis_r1", size, "-hip_step_graphs 1 -hip_fuse_steps 0", init)
    g2.run_solution(0, 8)
    st = g2.get_stats()
    assert st.get_num_fused_passes() == 0 and st.get_num_graph_steps() == 8
    ref = O.run_axis3(size, 9, radius=1)[("A", 9)]
    assert O.rel_linf(whole(g2, "A", 9), ref) <= 1e-12


def test_reverse_direction(gpu):
    """run_solution(hi, lo) on the reverse-time test stencil: the captured chain walks the step indices downwards."""
    from yask_amd import yk_factory

    def run(opts):
        fac = yk_factory("test_reverse_2d")
        s = fac.new_solution(fac.new_env())
        s.set_overall_domain_size_vec([40, 44])
        assert s.apply_command_line_options("-no-auto_tune " + opts) == ""
        s.prepare_solution()
        for i, v in enumerate(s.get_vars()):
            v.set_elements_hash(0.0, 1.0, hash_id=i)
        s.run_solution(0, -7)
        st = s.get_stats()
        out = []
        for v in s.get_vars():
            if v.get_dim_names()[:1] == ["t"] and v.get_num_dims() == 3:
                for t in (v.get_first_valid_step_index(), v.get_last_valid_step_index()):
                    out.append((t, v.get_elements_in_slice([t, 0, 0], [t, 39, 43])[0]))
        return out, st

    (a, sa), (b, sb) = run("-hip_step_graphs 1"), run("-hip_step_graphs 0")
    assert sa.get_num_graph_steps() == 8 and sb.get_num_graph_steps() == 0 and len(a) > 0
    for (tx, x), (ty, y) in zip(a, b):
        assert tx == ty and np.array_equal(x, y)


@pytest.mark.parametrize("stencil,size", [("test_step_cond_1d", [64]), ("swe2d", [48, 40])])
def test_solutions_whose_launches_depend_on_the_step_index_are_never_replayed(gpu, stencil, size):
    from yask_amd import yk_factory
    fac = yk_factory(stencil)
    s = fac.new_solution(fac.new_env())
    s.set_overall_domain_size_vec(size)
    assert s.apply_command_line_options("-no-auto_tune -hip_step_graphs 1") == ""
    s.prepare_solution()
    s.run_solution(0, 7)
    st = s.get_stats()
    assert st.get_num_steps_done() == 8 and st.get_num_graph_steps() == 0
