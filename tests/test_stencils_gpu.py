"""GPU parity of the other two hot-path stencils against the oracle
and the golden outputs of the unmodified reference:
  3axis  (AxisStencil r=4, fp64): stated tolerance rel-Linf <= 1e-12 (fp64; averaging stencil, no growth);
  ssg    (staggered-grid elastic, fp32, 2 stages, 9 in-place fields): max|gpu-ref| / max|ref| <= 2e-5
         per field after the run.
"""
import json
from pathlib import Path

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden"
INDEX = json.load(open(G / "index.json"))


def make(stencil, size, opts=""):
    from yask_amd import yk_factory
    fac = yk_factory(stencil)
    env = fac.new_env()
    soln = fac.new_solution(env)
    soln.set_overall_domain_size_vec(list(size))
    if opts:
        assert soln.apply_command_line_options(opts) == ""
    soln.prepare_solution()
    init = O.DEFAULT_INIT[stencil]
    for v in soln.get_vars():
        v.set_elements_hash(*init[v.get_name()], hash_id=O.VAR_IDS[stencil][v.get_name()])
    return soln


def domain_slice(soln, var, t):
    n = soln.get_overall_domain_size_vec()
    has_t = var.get_num_dims() == 4
    first = ([t] if has_t else []) + [0, 0, 0]
    last = ([t] if has_t else []) + [x - 1 for x in n]
    a = var.get_elements_in_slice(first, last)
    return a[0] if has_t else a


def variant_names(stencil, part=0):
    from yask_amd import yk_factory
    fac = yk_factory(stencil)
    s = fac.new_solution(fac.new_env())
    return [n for n in s.get_kernel_variant_names(part) if not n.startswith("abl")]


# ------------------------------------------------------------------ 3axis fp64
def test_axis3_every_variant_matches_oracle(gpu):
    size, steps = (21, 35, 70), 3          # ragged vs every tile; z not a multiple of the 2-wide vector
    ref = O.run_axis3(size, steps)
    for name in variant_names("3axis"):
        soln = make("3axis", size, f"-hip_variant {name}")
        assert soln.get_element_bytes() == 8
        soln.run_solution(0, steps - 1)
        A = soln.get_var("A")
        for t in (steps - 1, steps):
            err = O.rel_linf(domain_slice(soln, A, t), ref[("A", t)])
            assert err <= 1e-12, (name, t, err)
        soln.end_solution()


def test_axis3_marching_shapes_agree_bit_for_bit_whatever_the_chunking(gpu):
    """fp64 twin of test_iso3dfd_gpu.test_marching_shapes_agree_bit_for_bit_whatever_the_chunking."""
    import re
    size, steps = (150, 45, 120), 3
    groups = {}
    for name in variant_names("3axis"):
        m = re.match(r"starlin_v\d+_z\d+_y\d+_r(\d+)_", name)
        if m:
            groups.setdefault(m.group(1), []).append(name)
    assert groups
    for ry, names in sorted(groups.items()):
        first = None
        for k, name in enumerate(names):
            soln = make("3axis", size, f"-hip_variant {name} -hip_xchunk {(37, 64, 0, 51)[k % 4]}")
            soln.run_solution(0, steps - 1)
            got = domain_slice(soln, soln.get_var("A"), steps).copy()
            soln.end_solution()
            if first is None:
                first = got
            else:
                assert np.array_equal(first, got), (ry, names[0], name, int((first != got).sum()))


@pytest.mark.parametrize("name", [n for n in INDEX if INDEX[n]["stencil"] == "3axis" and "lattice_stride" not in INDEX[n] and INDEX[n].get("radius", 4) == 4])
def test_axis3_matches_reference_golden(gpu, name):
    meta = INDEX[name]
    z = np.load(G / f"{name}.npz")
    soln = make("3axis", meta["size"])
    soln.run_solution(0, meta["steps"] - 1)
    A = soln.get_var("A")
    for t in (meta["steps"] - 1, meta["steps"]):
        assert O.rel_linf(domain_slice(soln, A, t), z[f"A@{t}"]) <= 1e-12


def test_axis3_constant_field_is_a_fixed_point(gpu):
    """Size-independent property: the update is an average, so a constant field stays constant
    (weights sum to 1 up to the 15-digit literal of 1/25)."""
    soln = make("3axis", (40, 64, 256))
    soln.get_var("A").set_all_elements_same(3.25)
    soln.run_solution(0, 4)
    got = domain_slice(soln, soln.get_var("A"), 5)
    assert np.abs(got - 3.25).max() <= 1e-13


def test_axis3_radius1_heat3d_every_variant_matches_oracle(gpu):
    """BASELINE config 3 calls the fp64 case "heat3d": the classic 7-point stencil is AxisStencil at radius 1."""
    size, steps = (19, 33, 66), 4
    ref = O.run_axis3(size, steps, radius=1)
    from yask_amd import yk_factory
    fac = yk_factory("3axis_r1")
    names = [n for n in fac.new_solution(fac.new_env()).get_kernel_variant_names(0) if not n.startswith("abl")]
    for name in names:
        soln = fac.new_solution(fac.new_env())
        soln.set_overall_domain_size_vec(list(size))
        assert soln.apply_command_line_options(f"-hip_variant {name}") == ""
        soln.prepare_solution()
        A = soln.get_var("A")
        assert A.get_left_halo_size("x") == 1 and A.get_right_halo_size("z") == 1
        A.set_elements_hash(0.0, 1.0, hash_id=0)
        soln.run_solution(0, steps - 1)
        err = O.rel_linf(domain_slice(soln, A, steps), ref[("A", steps)])
        assert err <= 1e-12, (name, err)


# ------------------------------------------------------------------ ssg fp32
def _ssg_err(soln, ref, steps):
    worst = 0.0
    for n in O.SSG_FIELDS:
        got = domain_slice(soln, soln.get_var(n), steps).astype(np.float64)
        r = ref[(n, steps)].astype(np.float64)
        worst = max(worst, float(np.abs(got - r).max()) / max(1e-30, float(np.abs(r).max())))
    return worst


def test_ssg_every_variant_matches_oracle(gpu):
    size, steps = (19, 22, 37), 2
    ref = O.run_ssg(size, steps)
    names0, names1 = variant_names("ssg", 0), variant_names("ssg", 1)
    for name in sorted(set(names0) | set(names1)):
        soln = make("ssg", size, f"-hip_variant {name}")
        soln.run_solution(0, steps - 1)
        err = _ssg_err(soln, ref, steps)
        assert err <= 2e-5, (name, err)
        soln.end_solution()


def test_ssg_fast_division_is_an_option_and_the_exact_shapes_are_bit_identical(gpu):
    """ssg's default shapes issue a / b as a * v_rcp_f32(b) (<= 1.5 ulp of the quotient; ykh_march.hpp `MarchAcc`, `_fd`), a - b as
    fma(b, -1, a) (`_ps`, exact) and rename their register queues inside trips (`_t2`, exact).  -no-hip_fast_div selects the
    correctly rounded siblings, whose results must be BIT-identical to the plain shape of round 1; the fast shapes stay within a
    few ulp of them (and inside the 2e-5 parity bound against the oracle, like every shape: test above)."""
    size, steps = (256, 256, 256), 3          # large enough for the static defaults (no one-off timing of small grids)
    fast = make("ssg", size)
    assert "_fd" in fast.get_kernel_variant(0) and "_fd" in fast.get_kernel_variant(1)
    exact = make("ssg", size, "-no-hip_fast_div")
    assert "_fd" not in exact.get_kernel_variant(0) + exact.get_kernel_variant(1) and "_ps_t2" in exact.get_kernel_variant(0)
    plain = make("ssg", size, "-hip_variant march_v4_z128_y16_nt_hr_w2")
    assert plain.get_kernel_variant(0) == plain.get_kernel_variant(1) == "march_v4_z128_y16_nt_hr_w2"
    for s in (fast, exact, plain):
        s.run_solution(0, steps - 1)
    worst = 0.0
    for n in O.SSG_FIELDS:
        f, e, p = (domain_slice(s, s.get_var(n), steps) for s in (fast, exact, plain))
        assert np.array_equal(e, p), n
        worst = max(worst, float(np.abs(f.astype(np.float64) - p).max()) / float(np.abs(p).max()))
    assert 0.0 < worst <= 1e-6, worst


def test_ssg_default_shapes_are_bit_identical_under_odd_x_chunks(gpu):
    """The march kernels evaluate the generated expression with every operation pinned in the reference's order (no free choice of
    which product to fuse): the x-chunk length, odd ones included, must not move a bit (cf. the starlin shapes' test above)."""
    size, steps = (150, 64, 128), 3
    runs = []
    for opts in ("-hip_variant march_v4_z128_y16_nt_hr_ps_fd_t2_w2", "-hip_variant march_v4_z128_y16_nt_hr_ps_fd_t2_w2 -hip_xchunk 37",
                 "-hip_variant march_v4_z128_y16_nt_hr_ps_fd_t2_w2 -hip_xchunk 51"):
        s = make("ssg", size, opts)
        s.run_solution(0, steps - 1)
        runs.append({n: domain_slice(s, s.get_var(n), steps).copy() for n in O.SSG_FIELDS})
        s.end_solution()
    for n in O.SSG_FIELDS:
        assert np.array_equal(runs[0][n], runs[1][n]) and np.array_equal(runs[0][n], runs[2][n]), n


@pytest.mark.parametrize("name", [n for n in INDEX if INDEX[n]["stencil"] == "ssg" and "lattice_stride" not in INDEX[n]])
def test_ssg_matches_reference_golden(gpu, name):
    meta = INDEX[name]
    z = np.load(G / f"{name}.npz")
    soln = make("ssg", meta["size"])
    st = meta["steps"]
    soln.run_solution(0, st - 1)
    for n in O.SSG_FIELDS:
        v = soln.get_var(n)
        assert v.get_first_valid_step_index() == st and v.get_last_valid_step_index() == st   # 1 slot, in place
        got = domain_slice(soln, v, st).astype(np.float64)
        r = z[f"{n}@{st}"].astype(np.float64)
        assert np.abs(got - r).max() / max(1e-30, np.abs(r).max()) <= 2e-5, n
    for n in O.SSG_COEFFS:
        assert np.array_equal(domain_slice(soln, soln.get_var(n), 0), z[f"{n}@0"])


def test_ssg_stats_and_stage_structure(gpu):
    soln = make("ssg", (16, 16, 32))
    soln.run_solution(0, 1)
    st = soln.get_stats()
    assert st.get_num_steps_done() == 2
    assert st.get_num_writes_done() == 2 * 9 * 16 * 16 * 32          # 3 + 6 writes per point-step
    assert st.get_est_fp_ops_done() == 2 * (129 + 158) * 16 * 16 * 32


# ------------------------------------------------------------------ wave-front temporal tiling (-Mbt / -bt)
@pytest.mark.parametrize("stencil,size,variant", [("iso3dfd", (44, 37, 70), "starlin_v4_z128_y16_r1_m_nt_w2_c4"),
                                                  ("3axis", (40, 36, 66), "starlin_v2_z64_y32_r2_m_nt_w2_c4"),
                                                  ("ssg", (36, 22, 40), "march_v2_z128_y8_w2")])
@pytest.mark.parametrize("opts,steps", [("-Mbt 2 -Mbx 16", 4), ("-Mbt 3 -Mbx 16", 7), ("-bt 4", 5)])
def test_wavefront_temporal_tiling_is_bit_identical_to_plain_sweeps(gpu, stencil, size, variant, opts, steps):
    """The reference's mega-block wave-fronts (context.cpp:482-745,1181-1525) at launch granularity: groups of -Mbt steps
    are applied slab by slab, each (step, stage) shifted by the x-halo.  Exact by construction: compared bit for bit with the
    plain schedule (same kernel shape), for step counts that are and are not multiples of the group, 1 and 2 stages."""
    a = make(stencil, size, f"-hip_variant {variant} {opts}")
    b = make(stencil, size, f"-hip_variant {variant}")
    a.run_solution(0, steps - 1)
    b.run_solution(0, steps - 1)
    assert a.compare_data(b, 0.0) == 0
    for v in a.get_vars():
        if a.get_step_dim_name() in v.get_dim_names():
            assert v.get_first_valid_step_index() == b.get_var(v.get_name()).get_first_valid_step_index()
    assert a.get_stats().get_num_steps_done() == steps
    ref = {"iso3dfd": O.run_iso3dfd, "3axis": O.run_axis3, "ssg": O.run_ssg}[stencil](size, steps)
    name = {"iso3dfd": "p", "3axis": "A", "ssg": "v_tr_u"}[stencil]
    got = domain_slice(a, a.get_var(name), steps).astype(np.float64)
    r = ref[(name, steps)].astype(np.float64)
    assert np.abs(got - r).max() / max(1e-30, np.abs(r).max()) <= 2e-5
