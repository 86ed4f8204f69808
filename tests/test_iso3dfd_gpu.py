"""GPU parity of the iso3dfd hot path (HIP kernels through the C ABI) against the oracle and the
golden reference outputs.  Stated tolerance (fp32): rel-Linf <= 2e-5 after the
run, i.e. max|gpu-ref| / max(1, max|ref|); the reference itself only requires 1e-3."""
import json
from pathlib import Path

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden"
TOL = 2e-5


def make(size, opts="", stencil="iso3dfd"):
    from yask_amd import yk_factory
    fac = yk_factory(stencil)
    env = fac.new_env()
    soln = fac.new_solution(env)
    soln.set_overall_domain_size_vec(list(size))
    if opts:
        assert soln.apply_command_line_options(opts) == ""
    soln.prepare_solution()
    init = O.DEFAULT_INIT[stencil]
    for i, v in enumerate(soln.get_vars()):
        v.set_elements_hash(*init[v.get_name()], hash_id=O.VAR_IDS[stencil][v.get_name()])
    return fac, env, soln


def domain_slice(soln, var, t):
    n = soln.get_overall_domain_size_vec()
    first = [t, 0, 0, 0] if var.get_num_dims() == 4 else [0, 0, 0]
    last = [t] + [x - 1 for x in n] if var.get_num_dims() == 4 else [x - 1 for x in n]
    a = var.get_elements_in_slice(first, last)
    return a[0] if var.get_num_dims() == 4 else a


def variants():
    from yask_amd import yk_factory
    fac = yk_factory("iso3dfd")
    env = fac.new_env()
    s = fac.new_solution(env)
    # 'abl*' variants are deliberately-wrong profiling ablations
    return [n for n in s.get_kernel_variant_names(0) if not n.startswith('abl')]


def test_every_kernel_variant_matches_oracle(gpu):
    size, steps = (40, 37, 70), 3        # not multiples of any tile, z not a multiple of 4
    ref = O.run_iso3dfd(size, steps)
    for name in variants():
        _, _, soln = make(size, f"-hip_variant {name}")
        assert soln.get_kernel_variant(0) == name
        soln.run_solution(0, steps - 1)
        p = soln.get_var("p")
        for t in (steps - 1, steps):
            err = O.rel_linf(domain_slice(soln, p, t), ref[("p", t)])
            assert err <= TOL, (name, t, err)
        soln.end_solution()


def test_marching_shapes_agree_bit_for_bit_whatever_the_chunking(gpu):
    """Every starlin shape with the same rows-per-thread count sums in the same order (x past, own rows, slab rows, z, x future):
    tile size, queue rotation (_m / _t / _t2 / _u), prefetch depth, cheap tails (_tl) and the x-chunk length -- odd ones included --
    must not move the last bit.  (They did until the sums were explicit FMAs: ykh_device.hpp fmacc.)"""
    import re
    size, steps = (150, 45, 200), 3
    groups = {}
    for name in variants():
        m = re.match(r"starlin_v\d+_z\d+_y\d+_r(\d+)_", name)
        if m:
            groups.setdefault(m.group(1), []).append(name)
    assert groups
    for ry, names in sorted(groups.items()):
        first = None
        for k, name in enumerate(names):
            _, _, soln = make(size, f"-hip_variant {name} -hip_xchunk {(37, 64, 0, 51)[k % 4]}")
            soln.run_solution(0, steps - 1)
            got = domain_slice(soln, soln.get_var("p"), steps).copy()
            soln.end_solution()
            if first is None:
                first = got
            else:
                assert np.array_equal(first, got), (ry, names[0], name, int((first != got).sum()))


@pytest.mark.parametrize("xchunk", [1, 5, 16, 1000])
def test_x_chunking_is_transparent(gpu, xchunk):
    size, steps = (33, 20, 64), 2
    ref = O.run_iso3dfd(size, steps)
    _, _, soln = make(size, f"-hip_xchunk {xchunk}")
    soln.run_solution(0, steps - 1)
    err = O.rel_linf(domain_slice(soln, soln.get_var("p"), steps), ref[("p", steps)])
    assert err <= TOL, err


@pytest.mark.parametrize("name", ["iso3dfd_32x24x40_s3", "iso3dfd_20x52x36_s5"])
def test_matches_reference_golden(gpu, name):
    meta = json.load(open(G / "index.json"))[name]
    z = np.load(G / f"{name}.npz")
    _, _, soln = make(meta["size"])
    soln.run_solution(0, meta["steps"] - 1)
    p = soln.get_var("p")
    assert p.get_first_valid_step_index() == meta["steps"] - 1
    assert p.get_last_valid_step_index() == meta["steps"]
    for t in (meta["steps"] - 1, meta["steps"]):
        got = domain_slice(soln, p, t)
        r = z[f"p@{t}"]
        assert O.rel_linf(got, r) <= TOL, (t, O.rel_linf(got, r))
        assert O.within_tolerance(got, r).all()       # the reference's own acceptance rule
    # the hash init reproduces the reference's inputs bit-exactly
    assert np.array_equal(domain_slice(soln, soln.get_var("v"), 0), z["v@0"])


def test_step_by_step_equals_one_call(gpu):
    size = (24, 24, 32)
    opt = "-hip_variant starlin_v4_z128_y16_r1_m_nt_w2_c4"     # same kernel in both (small grids are auto-selected by timing)
    _, _, a = make(size, opt)
    _, _, b = make(size, opt)
    a.run_solution(0, 5)
    for t in range(6):
        b.run_solution(t)
    assert a.compare_data(b, 0.0) == 0
    st = a.get_stats()
    assert st.get_num_steps_done() == 6 and st.get_num_elements() == 24 * 24 * 32
    assert st.get_num_writes_done() == 6 * 24 * 24 * 32
    assert st.get_est_fp_ops_done() == 61 * 6 * 24 * 24 * 32


def test_linearity_at_full_width_rows(gpu):
    """Size-independent property: the update is linear in p for fixed v. Uses a 256-wide z so that
    the widest tiles are exercised: run(p1)+run(p2) == run(p1+p2) up to fp32 rounding."""
    size, steps = (20, 40, 256), 2
    outs = []
    for mode in range(3):
        _, _, soln = make(size)
        p = soln.get_var("p")
        if mode == 0:
            p.set_elements_hash(0.0, 1.0, hash_id=7)
        elif mode == 1:
            p.set_elements_hash(0.0, 1.0, hash_id=9)
        else:
            # p1 + p2 assembled on the host over domain+halo for both slots
            h = 8
            for t in (0, 1):
                f = [t, -h, -h, -h]
                l = [t, size[0] + h - 1, size[1] + h - 1, size[2] + h - 1]
                _, _, s1 = make(size); s1.get_var("p").set_elements_hash(0.0, 1.0, hash_id=7)
                _, _, s2 = make(size); s2.get_var("p").set_elements_hash(0.0, 1.0, hash_id=9)
                a1 = s1.get_var("p").get_elements_in_slice(f, l)
                a2 = s2.get_var("p").get_elements_in_slice(f, l)
                p.set_elements_in_slice(a1 + a2, f, l)
        soln.run_solution(0, steps - 1)
        outs.append(domain_slice(soln, p, steps).astype(np.float64))
    err = np.abs(outs[0] + outs[1] - outs[2]).max() / max(1.0, np.abs(outs[2]).max())
    assert err <= 1e-5, err


def test_full_size_1024_properties(gpu):
    """BASELINE.json configs[1] size (1024^3, ~15 GB per solution), size-independent properties:
    (a) the tuned kernel agrees with the generic point kernel everywhere (reference rule, eps 1e-4 << 1e-3);
    (b) a constant wavefield is a fixed point for any velocity model (the FD weights sum to zero)."""
    n, steps = 1024, 2
    _, _, a = make((n, n, n))
    assert a.get_kernel_variant(0).startswith("starlin")
    _, _, b = make((n, n, n), "-force_scalar")
    assert b.get_kernel_variant(0) == "naive"
    a.run_solution(0, steps - 1)
    b.run_solution(0, steps - 1)
    assert a.compare_data(b, 1e-4) == 0
    b.end_solution()
    p = a.get_var("p")
    p.set_all_elements_same(0.75)
    a.run_solution(steps, steps + 2)
    r = p.reduce_elements_in_slice(8 | 16, [steps + 3, 0, 0, 0], [steps + 3, n - 1, n - 1, n - 1])
    assert abs(r.get_max() - 0.75) <= 2e-6 and abs(r.get_min() - 0.75) <= 2e-6
    assert r.get_num_elements_reduced() == n ** 3
    a.end_solution()


def test_more_tiles_than_cus(gpu):
    """A plane of 9 x 32 = 288 tiles on 256 CUs: the tile rows are launched one CU-filling round at a time; the result equals the
    point kernel's."""
    size, steps = (12, 1024, 1100), 2
    _, _, a = make(size, "")
    assert a.get_kernel_variant(0).startswith("starlin")
    _, _, b = make(size, "-force_scalar")
    a.run_solution(0, steps - 1)
    b.run_solution(0, steps - 1)
    assert a.compare_data(b, 1e-4) == 0
    a.end_solution()
    b.end_solution()


def test_auto_tuner_picks_a_variant_and_preserves_data(gpu):
    """run_auto_tuner_now() (yk_solution_api.hpp:880) times the compiled tile shapes on the real vars and must leave
    their contents untouched; the solution then runs with the chosen shape and still matches the oracle."""
    size, steps = (64, 48, 160), 3
    ref = O.run_iso3dfd(size, steps)
    _, _, soln = make(size)
    before = domain_slice(soln, soln.get_var("p"), 1).copy()
    soln.run_auto_tuner_now(False)
    assert np.array_equal(domain_slice(soln, soln.get_var("p"), 1), before)
    chosen = soln.get_kernel_variant(0)
    # (on a grid this small any family may win, the point kernel included)
    assert chosen in soln.get_kernel_variant_names(0) and not chosen.startswith("abl")
    soln.run_solution(0, steps - 1)
    assert O.rel_linf(domain_slice(soln, soln.get_var("p"), steps), ref[("p", steps)]) <= TOL
    soln.end_solution()
