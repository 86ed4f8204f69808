"""Two time steps per pass, fused on chip (yask_amd/csrc/ykh_starlin2.hpp, option -hip_fuse_steps 2): the GPU
counterpart of the reference's temporal blocking (`-bt`, src/kernel/lib/context.cpp:747-819) for the stencils whose
two-step state fits on chip -- the AxisStencil family (`3axis`, radius 4 and radius 1; SimpleStencils.cpp:61-103).
S(t+1) never goes to memory except on the last pass.  Checked against the plain one-step schedule (same arithmetic
per point: <= 1e-13), against the oracle (<= 1e-12), for even / odd step counts (the odd last step runs the plain
kernel), sizes that are not multiples of any tile, forced x-chunks, and with the per-slot hash init whose pads differ
between the two step slots (step t+2 at the boundary must see the t+1 slot's pads, exactly as a plain run does)."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
RADIUS = {"3axis": 4, "3axis_r1": 1}


def make(stencil, size, opts=""):
    from yask_amd import yk_factory
    fac = yk_factory(stencil)
    soln = fac.new_solution(fac.new_env())
    soln.set_overall_domain_size_vec(list(size))
    assert soln.apply_command_line_options("-no-auto_tune " + opts) == ""
    soln.prepare_solution()
    soln.get_var("A").set_elements_hash(0.0, 1.0, hash_id=0)
    return soln


def field(soln, t):
    n = soln.get_overall_domain_size_vec()
    return soln.get_var("A").get_elements_in_slice([t, 0, 0, 0], [t] + [x - 1 for x in n])[0]


@pytest.mark.parametrize("stencil", ["3axis", "3axis_r1"])
@pytest.mark.parametrize("size,steps,extra", [((40, 61, 130), 4, ""), ((40, 61, 130), 5, ""), ((96, 100, 200), 2, ""),
                                              ((30, 24, 56), 6, ""), ((70, 90, 140), 4, "-hip_xchunk 16"), ((33, 50, 70), 1, "")])
def test_two_steps_per_pass_equal_plain_sweeps(gpu, stencil, size, steps, extra):
    fused = make(stencil, size, "-hip_fuse_steps 2 " + extra)
    plain = make(stencil, size, "-hip_fuse_steps 0")
    fused.run_solution(0, steps - 1)
    plain.run_solution(0, steps - 1)
    A = fused.get_var("A")
    assert A.get_last_valid_step_index() == steps and A.get_first_valid_step_index() == steps - 1
    ref = O.run_axis3(size, steps, radius=RADIUS[stencil])
    for t in (steps - 1, steps):         # BOTH step slots hold what a plain run leaves there
        f, p = field(fused, t), field(plain, t)
        assert np.abs(f - p).max() <= 1e-13 * max(1.0, np.abs(p).max()), (t, np.abs(f - p).max())
        assert O.rel_linf(f, ref[("A", t)]) <= 1e-12, t
    st = fused.get_stats()
    assert st.get_num_steps_done() == steps and st.get_num_fused_passes() == steps // 2
    # pads of both slots are untouched (a single rank never writes its pads)
    h = RADIUS[stencil]
    for t in (steps - 1, steps):
        for sl in ([t, -h, 0, 0], [t, -1, size[1] - 1, size[2] - 1]), ([t, 0, 0, size[2]], [t, size[0] - 1, size[1] - 1, size[2] + h - 1]):
            a = fused.get_var("A").get_elements_in_slice(*sl)
            b = plain.get_var("A").get_elements_in_slice(*sl)
            assert np.array_equal(a, b)


def test_fused_runs_can_be_continued_and_mixed(gpu):
    """run_solution() in several calls, fused and plain mixed: the step slots are consistent after every call."""
    size = (48, 56, 120)
    fused = make("3axis", size, "-hip_fuse_steps 2")
    plain = make("3axis", size, "-hip_fuse_steps 0")
    t = 0
    for n in (2, 3, 4, 1, 6):
        fused.run_solution(t, t + n - 1)
        plain.run_solution(t, t + n - 1)
        t += n
        for s in (t - 1, t):
            f, p = field(fused, s), field(plain, s)
            assert np.abs(f - p).max() <= 1e-13 * max(1.0, np.abs(p).max()), (t, s)


def test_fusion_is_refused_where_it_does_not_apply(gpu):
    """iso3dfd has no two-step kernel (r=8 does not fit): the option is accepted and plain sweeps run."""
    from yask_amd import yk_factory
    fac = yk_factory("iso3dfd")
    s = fac.new_solution(fac.new_env())
    s.set_overall_domain_size_vec([32, 32, 64])
    assert s.apply_command_line_options("-hip_fuse_steps 2") == ""
    s.prepare_solution()
    s.get_var("p").set_elements_hash(0.0, 1.0, hash_id=0)
    s.get_var("v").set_elements_hash(150.0, 50.0, hash_id=1)
    s.run_solution(0, 3)
    ref = O.run_iso3dfd((32, 32, 64), 4)[("p", 4)]
    got = s.get_var("p").get_elements_in_slice([4, 0, 0, 0], [4, 31, 31, 63])[0]
    assert O.rel_linf(got, ref) <= 2e-5


def test_fusion_is_opt_in(gpu):
    """Default (-hip_fuse_steps not given): plain sweeps -- run_solution(a, b) is then bit-identical to one call per step
    (the fused pass contracts its FMAs differently, so it must not be what an unsuspecting caller gets).
    -hip_fuse_steps 2 switches the two-steps-per-pass kernel on where the solution has one."""
    for stencil, opts, want in (("3axis_r1", "", False), ("3axis", "", False), ("3axis_r1", "-hip_fuse_steps 2", True),
                                ("3axis_r1", "-hip_fuse_steps 0", False), ("3axis", "-hip_fuse_steps 2", True)):
        s = make(stencil, (40, 40, 72), opts)
        assert ("-hip_fuse_steps" in s.get_command_line_values()) == want
        s.run_solution(0, 3)
        p = make(stencil, (40, 40, 72), "")
        for t in range(4):
            p.run_solution(t, t)
        if want:
            assert np.abs(field(s, 4) - field(p, 4)).max() <= 1e-13
        else:
            assert np.array_equal(field(s, 4), field(p, 4))          # one call over four steps == four calls, bit for bit
        assert s.get_stats().get_num_fused_passes() == (2 if want else 0) and p.get_stats().get_num_fused_passes() == 0
