"""The generic kernel families against the REFERENCE on grids of several tiles.

tests/test_reference_stencils_gpu.py holds every registered shape of every solution to outputs of the unmodified reference -- on
20 x 18 x 24-class grids, where a workgroup's tile covers the whole (y, z) plane and one x-chunk the whole x range.  What the round-5
kernels added (the plane-ring `box_kernel`, the equation clusters `c<K>_*`, the box-list dispatch of sub-domain parts, the 32 x 32 and
lanes-along-y point tiles) is only non-trivial on grids of several tiles and several x-chunks -- ring wrap across chunk seams, tile
halos that belong to a neighbouring tile, prefetch past the chunk, shell boxes thin in z -- and there the only checker used to be
another HIP kernel (tests/test_box_kernel_gpu.py, test_clusters_gpu.py, test_part_boxes_gpu.py).

Here: every 3-D generic solution on a ragged 136 x 72 x 264 grid against lattice samples of the reference's own result at that size
(tests/golden/*_mt.npz, made by tests/golden/make_golden.py TILE_CASES from oracle/_ref: all points of the 9-wide boundary layers,
every 16th point, both sides of every multiple of 32 per dim).  The timed default AND every registered shape of every part, each also
with a forced 48-plane x-chunk (seams at x = 48 and 96, both on the lattice).  `cube` and `fsg_abc` also cut over 2 and 8 ranks,
against the fixture (not against the one-rank run).  The reference validates these solutions at its default sizes, not at one vector
(src/kernel/Makefile:1130-1166).

Tolerance (fp32): max|gpu - ref| / max|ref| <= 2e-5 per array -- the bound of the one-tile tests; the reference's AVX-512 kernel and
these kernels associate the sums differently."""
import json
import os
import socket
from pathlib import Path

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden"
INDEX = json.load(open(G / "index.json"))
CASES = sorted(n for n in INDEX if INDEX[n].get("multi_tile"))
TOL = 2e-5


def _lattice_of(soln, name, t, size, lat):
    """the var's values on the fixture's lattice: whole x-planes are fetched (the lattice's x indices only), sampled in y and z"""
    var = soln.get_var(name)
    ix, iy, iz = (O.lattice(n, **lat) for n in size)
    # trailing misc dims (ssg2 / fsg2: v(t, x, y, z, vidx)) are kept whole
    misc = [d for d in var.get_dim_names()[4:]]
    m0 = [var.get_first_misc_index(d) for d in misc]
    m1 = [var.get_last_misc_index(d) for d in misc]
    mshape = tuple(b - a + 1 for a, b in zip(m0, m1))
    out = np.empty((len(ix), len(iy), len(iz)) + mshape, dtype=np.float64)
    for k, x in enumerate(ix):
        pl = np.asarray(var.get_elements_in_slice([t, int(x), 0, 0] + m0, [t, int(x), size[1] - 1, size[2] - 1] + m1))
        out[k] = pl.reshape((size[1], size[2]) + mshape)[iy][:, iz]
    return out


def _prepared(fac, env, meta, opts="", nr=None):
    s = fac.new_solution(env)
    s.set_overall_domain_size_vec(meta["size"])
    if nr is not None:
        s.set_num_ranks_vec(list(nr))
    if opts:
        assert s.apply_command_line_options(opts) == "", opts
    s.prepare_solution()
    for i, v in enumerate(s.get_vars()):
        v.set_elements_hash(*meta["init"], hash_id=i)
    return s


def _compare(soln, meta, z, what):
    worst = 0.0
    for key in meta["arrays"]:
        vname, t = key.split("@")
        ref = z[key].astype(np.float64)
        assert np.isfinite(ref).all()
        got = _lattice_of(soln, vname, int(t), meta["size"], meta["lattice"])
        assert got.shape == ref.shape, (key, got.shape, ref.shape)
        err = np.abs(got - ref).max() / max(1e-30, np.abs(ref).max())
        assert err <= TOL, (what, key, err, np.unravel_index(np.abs(got - ref).argmax(), ref.shape))
        worst = max(worst, err)
    return worst


def test_fixture_set_is_complete():
    """every 3-D solution the generic registry serves has a multi-tile fixture (the three hot-path stencils have their own, larger ones)"""
    have = {INDEX[n]["stencil"] for n in CASES}
    want = {"cube", "3plane", "3axis_with_diags", "tti", "fsg", "fsg_abc", "fsg2_abc", "awp", "awp_abc", "awp_elastic_abc",
            "iso3dfd_sponge", "ssg2", "test_3d", "test_boundary_3d", "test_scratch_3d", "test_stages_3d", "test_partial_3d", "test_stream_3d"}
    assert want <= have, want - have
    for n in CASES:
        m = INDEX[n]
        assert m["size"] == [136, 72, 264] and m["lattice"] == {"stride": 16, "edge": 9, "tile": 32} and m["arrays"], n


@pytest.mark.parametrize("name", CASES)
def test_every_registered_shape_matches_the_reference_on_a_multi_tile_grid(gpu, name):
    from yask_amd import yk_factory
    meta = INDEX[name]
    z = np.load(G / f"{name}.npz")
    fac = yk_factory(meta["stencil"])
    env = fac.new_env()
    last = meta["steps"] - 1
    # 1. what prepare_solution() picks by timing on this grid (kernel family, x-chunk), i.e. what a user gets
    s = _prepared(fac, env, meta)
    s.run_solution(0, last)
    chosen = [s.get_kernel_variant(p) for p in range(s.get_num_parts())]
    _compare(s, meta, z, ("default", chosen))
    names = []
    for part in range(s.get_num_parts()):
        for vn in s.get_kernel_variant_names(part):
            if not vn.startswith("abl") and vn not in names:
                names.append(vn)
    families = {vn.split("_")[0] for vn in names}
    s.end_solution()
    # 2. every registered shape (a name applies to every part that has it; the other parts keep their timed choice), with the
    #    library's own x-chunking and with seams forced at x = 48 and 96
    ran = 0
    for vn in names:
        for xo in ("", " -hip_xchunk 48"):
            s2 = _prepared(fac, env, meta, f"-hip_variant {vn}{xo}")
            assert any(s2.get_kernel_variant(p) == vn for p in range(s2.get_num_parts())), vn
            s2.run_solution(0, last)
            _compare(s2, meta, z, (vn, xo))
            s2.end_solution()
            ran += 1
    assert ran == 2 * len(names) and ran >= 2
    print(f"{name}: default {chosen}; {len(names)} shapes x 2 x-chunkings, families {sorted(families)}")


def test_the_round_5_families_are_among_the_shapes_held_to_the_reference(gpu):
    """the point of these fixtures: plane-ring box kernels, equation clusters, marching kernels and box-list parts are REGISTERED for the
    solutions above (so the parametrised test ran them) -- a registry change that drops them must not pass silently"""
    from yask_amd import yk_factory
    seen = {}
    for stencil in ("cube", "3plane", "tti", "fsg", "fsg_abc", "awp", "iso3dfd_sponge", "test_boundary_3d"):
        fac = yk_factory(stencil)
        s = fac.new_solution(fac.new_env())
        s.set_overall_domain_size_vec([136, 72, 264])
        s.prepare_solution()
        fam = set()
        for p in range(s.get_num_parts()):
            fam |= {vn.split("_")[0] for vn in s.get_kernel_variant_names(p)}
        seen[stencil] = fam
        if stencil in ("fsg_abc", "test_boundary_3d"):
            assert any(len(s.get_part_full_boxes(p)) >= 2 for p in range(s.get_num_parts())), stencil     # the shell: a list of boxes
        s.end_solution()
    assert "box" in seen["cube"] and "box" in seen["3plane"] and "box" in seen["tti"], seen
    assert any(f.startswith("c") and f[1:].isdigit() for f in seen["fsg"]), seen["fsg"]
    assert "march" in seen["awp"] or "starlin" in seen["awp"], seen["awp"]
    assert seen["iso3dfd_sponge"] & {"starlin", "march"}, seen["iso3dfd_sponge"]


# ------------------------------------------------------------------ cut over ranks, against the FIXTURE
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_worker(rank, world, port, name, nr, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      YASK_HIP_TRANSPORT="ipc", YASK_HIP_WAIT_TIMEOUT_S="30", HSA_ENABLE_IPC_MODE_LEGACY="0")
    from yask_amd import yk_factory
    meta = INDEX[name]
    fac = yk_factory(meta["stencil"])
    env = fac.new_env()
    env.init_from_launcher()
    s = _prepared(fac, env, meta, nr=nr)
    s.run_solution(0, meta["steps"] - 1)
    f, l = s.get_first_rank_domain_index_vec(), s.get_last_rank_domain_index_vec()
    t = meta["steps"]
    out = {}
    for key in meta["arrays"]:
        vname = key.split("@")[0]
        # this rank's share of the fixture's lattice
        idx = [O.lattice(n, **meta["lattice"]) for n in meta["size"]]
        mine = [ix[(ix >= f[d]) & (ix <= l[d])] for d, ix in enumerate(idx)]
        var = s.get_var(vname)
        # (trailing misc dims -- ssg2: v(t, x, y, z, vidx) -- are kept whole, as in _lattice_of)
        misc = [d for d in var.get_dim_names()[4:]]
        m0 = [var.get_first_misc_index(d) for d in misc]
        m1 = [var.get_last_misc_index(d) for d in misc]
        mshape = tuple(b - a_ + 1 for a_, b in zip(m0, m1))
        a = np.empty(tuple(len(m) for m in mine) + mshape, dtype=np.float64)
        for k, x in enumerate(mine[0]):
            pl = np.asarray(var.get_elements_in_slice([t, int(x), f[1], f[2]] + m0, [t, int(x), l[1], l[2]] + m1)).reshape((l[1] - f[1] + 1, l[2] - f[2] + 1) + mshape)
            a[k] = pl[mine[1] - f[1]][:, mine[2] - f[2]]
        out[key] = (mine, a)
    q.put((rank, out, [s.get_kernel_variant(p) for p in range(s.get_num_parts())]))
    env.global_barrier()
    s.end_solution()


@pytest.mark.parametrize("world,nr", [(2, (2, 1, 1)), (8, (2, 2, 2))], ids=["2ranks", "8ranks"])
@pytest.mark.parametrize("stencil", ["cube", "fsg_abc", "awp", "ssg2", "iso3dfd_sponge", "tti"])
def test_decomposed_runs_match_the_reference_fixture(gpu, stencil, world, nr):
    """N ranks on one device (IPC transport): the assembled lattice equals the REFERENCE's one-rank result at that size -- the checker
    is the fixture, not this library's own one-rank run (which tests/test_part_boxes_gpu.py and test_clusters_gpu.py compare bit for bit).
    awp / ssg2 / iso3dfd_sponge (end of round 6): one-part stages on generic marching shapes, which now have descriptor-reading twins --
    over 2 x 2 x 2 ranks they run the planned / halves schedules instead of exterior slabs + interior; likewise cube and tti on the
    plane-ring kernel's twin (planned launches: their vars are read at mixed offsets, which rules the halves out)."""
    import multiprocessing as mp
    name = [n for n in CASES if INDEX[n]["stencil"] == stencil][0]
    meta = INDEX[name]
    z = np.load(G / f"{name}.npz")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_worker, args=(r, world, port, name, nr, q)) for r in range(world)]
    for p in procs:
        p.start()
    parts = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    idx = [O.lattice(n, **meta["lattice"]) for n in meta["size"]]
    for key in meta["arrays"]:
        ref = z[key].astype(np.float64)
        got = np.full(ref.shape, np.nan)
        for _, out, _ in parts:
            mine, a = out[key]
            pos = [np.searchsorted(idx[d], mine[d]) for d in range(3)]
            got[np.ix_(*pos)] = a
        assert np.isfinite(got).all(), "the ranks' shares cover the lattice"
        err = np.abs(got - ref).max() / max(1e-30, np.abs(ref).max())
        assert err <= TOL, (key, err)
