"""The reference test matrix's COMPILE-TIME variants (VERDICT r05 next #3 / missing #2).

"Stencil definitions in src/stencils drop in unchanged" includes the options they are compiled with: the reference's `stencil-tests`
build several solutions with `radius=` and `domain_dims=` set (src/kernel/Makefile:1116-1153):

    iso3dfd radius=3 domain_dims=z,x,y      iso3dfd_sponge radius=6          test_stream_3d radius=5
    test_3d domain_dims=z,y,x               test_stages_3d domain_dims=x,z,y  test_partial_3d domain_dims=x,z,y
    test_2d domain_dims=y,x                 test_reverse_2d radius=1

Each is one more kernel library `libyask_kernel.<stencil><suffix>.cdna4_hip.so` (the reference's YK_STENCIL_SUFFIX), rendered by the
`cdna4_hip` target from the unchanged DSL source with those flags (yask_amd/csrc/variants.mk).  `-domain-dims` changes which dim is
outermost (the marching dim of the kernels here) and which is unit-stride (the lanes), while every var keeps its DECLARED dim order
(`p(t, x, y, z)` stays `t, x, y, z` for the API, ykh_var.cpp maps var positions to solution dims) -- so these cases permute exactly what
the kernels assume.  Goldens: the reference kernel built with the same flags (oracle/Makefile YC_EXTRA, tests/golden/make_golden.py
VARIANT_CASES), whole arrays.  Tolerance as for every generic solution: max|gpu - ref| / max|ref| <= 2e-5."""
import json
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden"
INDEX = json.load(open(G / "index.json"))
CASES = sorted(n for n in INDEX if INDEX[n].get("variant"))
# which kernel families must be REGISTERED for the variant's main part: a permuted star still reaches the linear-star kernel, a
# permuted general 3-D part the marching / vector kernels -- not silently the scalar point kernel alone
FAMILIES = {"iso3dfd-r3zxy": {"starlin"}, "iso3dfd_sponge-r6": {"starlin"}, "test_stream_3d-r5": {"vecpt"}, "test_3d-zyx": {"vecpt", "march"},
            "test_stages_3d-xzy": {"vecpt"}, "test_partial_3d-xzy": {"vecpt"}, "test_2d-yx": {"naive"}, "test_reverse_2d-r1": {"naive"}}
DIMS = {"iso3dfd-r3zxy": ["z", "x", "y"], "test_3d-zyx": ["z", "y", "x"], "test_stages_3d-xzy": ["x", "z", "y"],
        "test_partial_3d-xzy": ["x", "z", "y"], "test_2d-yx": ["y", "x"]}


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


def _run(fac, meta, opts=""):
    s = fac.new_solution(fac.new_env())
    s.set_overall_domain_size_vec(meta["size"])
    if opts:
        assert s.apply_command_line_options(opts) == ""
    s.prepare_solution()
    for i, v in enumerate(s.get_vars()):
        v.set_elements_hash(*meta["init"], hash_id=i)
    last = -(meta["steps"] - 1) if meta.get("reverse") else meta["steps"] - 1
    s.run_solution(0, last)
    return s


def _check(s, meta, z, what):
    for key in meta["arrays"]:
        vname, t = key.split("@")
        got = np.asarray(_slice(s, s.get_var(vname), int(t)), dtype=np.float64)
        ref = z[key].astype(np.float64)
        assert got.shape == ref.shape, (what, key, got.shape, ref.shape)
        assert np.isfinite(ref).all()
        err = np.abs(got - ref).max() / max(1e-30, np.abs(ref).max())
        assert err <= 2e-5, (what, key, err)


def test_all_eight_variants_of_the_reference_matrix_have_a_fixture():
    assert {INDEX[n]["stencil"] for n in CASES} == set(FAMILIES), CASES
    mk = (Path(__file__).resolve().parents[1] / "yask_amd" / "csrc" / "variants.mk").read_text()
    for n in CASES:
        m = INDEX[n]
        assert f"VF_{m['stencil']}" in mk and m["compiler_flags"] in mk, n       # library and golden were compiled with the same flags


@pytest.mark.parametrize("name", CASES)
def test_variant_matches_the_reference_built_with_the_same_flags(gpu, name):
    from yask_amd import yk_factory
    meta = INDEX[name]
    tag = meta["stencil"]
    z = np.load(G / f"{name}.npz")
    fac = yk_factory(tag)
    s = _run(fac, meta)
    assert s.get_name() == meta["solution"]
    if tag in DIMS:          # the solution's domain dims are the ones -domain-dims names, in that order; vars keep their declared order
        assert s.get_domain_dim_names() == DIMS[tag], s.get_domain_dim_names()
        full = [v for v in s.get_vars() if len(v.get_dim_names()) == len(DIMS[tag]) + 1]
        assert full and all(v.get_dim_names()[1:] == sorted(DIMS[tag]) for v in full), [v.get_dim_names() for v in full]
    _check(s, meta, z, "default")
    names, fams = [], set()
    for part in range(s.get_num_parts()):
        for vn in s.get_kernel_variant_names(part):
            fams.add(vn.split("_")[0])
            if not vn.startswith("abl") and vn not in names:
                names.append(vn)
    assert FAMILIES[tag] <= fams, (tag, fams)
    s.end_solution()
    for vn in names:             # every registered shape, not only the timed choice
        s2 = _run(fac, meta, f"-hip_variant {vn}")
        assert any(s2.get_kernel_variant(p) == vn for p in range(s2.get_num_parts())), vn
        _check(s2, meta, z, vn)
        s2.end_solution()


def test_permuted_star_runs_the_linear_star_kernel_at_size(gpu):
    """iso3dfd radius 3 with domain dims z,x,y on a grid large enough for the timing to matter: prepare_solution() must land on a
    marching family (the y dim is unit-stride here, z the marching dim), and two ranks cut along the outermost dim's neighbour equal one."""
    from yask_amd import yk_factory
    fac = yk_factory("iso3dfd-r3zxy")
    s = fac.new_solution(fac.new_env())
    s.set_overall_domain_size_vec([256, 256, 256])
    s.prepare_solution()
    assert s.get_kernel_variant(0).split("_")[0] in ("starlin", "march"), s.get_kernel_variant(0)
    p = s.get_var("p")
    assert p.get_dim_names() == ["t", "x", "y", "z"]
    # radius 3: halos of 3 in every domain dim, by NAME
    assert [p.get_left_halo_size(d) for d in ("x", "y", "z")] == [3, 3, 3]
    s.end_solution()


# ------------------------------------------------------------------ a permuted solution cut over ranks
def _rank_worker(rank, world, port, name, nr, q):
    import os
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      YASK_HIP_TRANSPORT="ipc", YASK_HIP_WAIT_TIMEOUT_S="30", HSA_ENABLE_IPC_MODE_LEGACY="0")
    from yask_amd import yk_factory
    meta = INDEX[name]
    fac = yk_factory(meta["stencil"])
    env = fac.new_env()
    env.init_from_launcher()
    s = fac.new_solution(env)
    s.set_overall_domain_size_vec(meta["size"])
    s.set_num_ranks_vec(list(nr))
    s.prepare_solution()
    for i, v in enumerate(s.get_vars()):
        v.set_elements_hash(*meta["init"], hash_id=i)
    s.run_solution(0, meta["steps"] - 1)
    dom = s.get_domain_dim_names()
    f = dict(zip(dom, s.get_first_rank_domain_index_vec()))
    l = dict(zip(dom, s.get_last_rank_domain_index_vec()))
    out = {}
    for key in meta["arrays"]:
        vname, t = key.split("@")
        var = s.get_var(vname)
        dn = var.get_dim_names()
        if len(dn) != len(dom) + 1 or dn[0] != s.get_step_dim_name() or int(t) != meta["steps"]:
            continue
        first = [int(t)] + [f[d] for d in dn[1:]]
        last = [int(t)] + [l[d] for d in dn[1:]]
        out[key] = (first[1:], last[1:], np.asarray(var.get_elements_in_slice(first, last))[0])
    q.put((rank, out))
    env.global_barrier()
    s.end_solution()


@pytest.mark.parametrize("tag,nr", [("iso3dfd-r3zxy", (2, 1, 1)), ("iso3dfd-r3zxy", (1, 1, 2)), ("test_3d-zyx", (1, 2, 1)), ("test_stages_3d-xzy", (2, 1, 2))],
                         ids=["iso3dfd-r3zxy-2x1x1", "iso3dfd-r3zxy-1x1x2", "test_3d-zyx-1x2x1", "test_stages_3d-xzy-2x1x2"])
def test_permuted_solution_cut_over_ranks_matches_the_reference(gpu, tag, nr):
    """the rank grid is given in the SOLUTION's domain-dim order (-domain-dims z,x,y: the first entry cuts z); every rank's box of the
    written var, addressed in the var's declared dim order, equals the reference's one-rank result built with the same flags"""
    import multiprocessing as mp
    import socket
    name = [n for n in CASES if INDEX[n]["stencil"] == tag][0]
    meta = INDEX[name]
    z = np.load(G / f"{name}.npz")
    world = int(np.prod(nr))
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_worker, args=(r, world, port, name, nr, q)) for r in range(world)]
    for p in procs:
        p.start()
    parts = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    checked = 0
    for key in meta["arrays"]:
        if not all(key in out for _, out in parts):
            continue
        ref = z[key].astype(np.float64)
        got = np.full(ref.shape, np.nan)
        for _, out in parts:
            first, last, a = out[key]
            got[tuple(slice(f0, l0 + 1) for f0, l0 in zip(first, last))] = a
        assert np.isfinite(got).all(), key
        err = np.abs(got - ref).max() / max(1e-30, np.abs(ref).max())
        assert err <= 2e-5, (key, err)
        checked += 1
    assert checked >= 1
