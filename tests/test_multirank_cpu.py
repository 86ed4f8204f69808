"""N>1 path on CPU (no GPU): the decomposition/halo-geometry code that prepare_solution() runs on every
GPU is exported device-free (yk_plan_rank / yk_plan_halo_slab, yask_amd/csrc/ykh_plan.cpp); here it drives
(a) rule checks against the reference's documented behaviour,
(b) an in-process emulation of a 2x2x2 rank grid, and
(c) a real 2-process `gloo` run: each rank owns a sub-box, exchanges the planned halo slabs with
    torch.distributed and steps its box with the oracle; the union must equal the single-rank oracle
    bit-for-bit (same arithmetic, same order, halos carry exactly the neighbour's values).
(b2)/(c2) the same two for the pipelined half-exchange schedule (-hip_halves): the box stepped in two x-halves, the faces cut by
    yk_plan_halves / yk_plan_halves_slab, a half's messages completed only after the next half has been computed;
(d)-(g) the native rendezvous and TCP mesh, the wave-front plan, the planned-launch block lists, the halves' block lists and cuts.
The data path on the GPUs (pack kernel -> RCCL -> unpack kernel) is covered by tests/test_multirank_gpu.py.
"""
import ctypes as C
import os
import socket

import numpy as np
import pytest

from oracle import oracle as O
from yask_amd import _capi

H = 8   # iso3dfd halo


def _lib():
    return _capi.load("iso3dfd")      # dlopen only: no device is touched


def plan(nranks, rank, global_size=None, local_size=None, num_ranks=(0, 0, 0)):
    p = _capi.RankPlan()
    for d in range(3):
        p.global_size[d] = (global_size or (0, 0, 0))[d]
        p.local_size[d] = (local_size or (0, 0, 0))[d]
        p.num_ranks[d] = num_ranks[d]
    rc = _lib().yk_plan_rank(3, nranks, rank, C.byref(p))
    if rc != 0:
        msg = _lib().yk_last_error().decode()
        _lib().yk_clear_error()
        raise RuntimeError(msg)
    return p


def slab(p, ofs, halo=(H, H, H), l1=1, sending=True):
    b = _capi.Box()
    o = (C.c_int * 3)(*ofs)
    hl = (_capi.idx_t * 3)(*halo)
    hr = (_capi.idx_t * 3)(*halo)
    rc = _lib().yk_plan_halo_slab(3, C.byref(p), o, hl, hr, l1, 1 if sending else 0, C.byref(b))
    assert rc >= 0
    return (tuple(b.first), tuple(b.size)) if rc == 1 else None


def neighbors(p):
    return [(p.neighbor_rank[i], tuple(p.neighbor_offset[i])) for i in range(p.num_neighbors)]


# ------------------------------------------------------------------ (a) rules
def test_compact_rank_grids_match_reference():
    # probed on the reference: 4 ranks -> x=2*y=2*z=1; 8 -> 2*2*2
    assert tuple(plan(4, 0, (64, 64, 64)).num_ranks) == (2, 2, 1)
    assert tuple(plan(8, 0, (64, 64, 64)).num_ranks) == (2, 2, 2)
    assert tuple(plan(2, 0, (64, 64, 64)).num_ranks)[::-1].count(1) == 2
    # a partly specified grid is completed; a contradictory one is re-factored without the pre-set
    # values, exactly as Tuple::get_compact_factors does (src/common/tuple.cpp:378-430: "keep", then not)
    assert tuple(plan(8, 0, (64, 64, 64), num_ranks=(8, 0, 0)).num_ranks) == (8, 1, 1)
    assert tuple(plan(8, 0, (64, 64, 64), num_ranks=(3, 1, 1)).num_ranks) == (2, 2, 2)
    with pytest.raises(RuntimeError, match="invalid rank"):
        plan(4, 4, (64, 64, 64))


def test_rank_coordinates_first_dim_fastest_and_remainder_on_last_rank():
    # API doc example: x varies fastest (include/aux/yk_solution_api.hpp:458-467)
    coords = [tuple(plan(6, r, (100, 90, 80), num_ranks=(2, 3, 1)).rank_index) for r in range(6)]
    assert coords == [(0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 0), (0, 2, 0), (1, 2, 0)]
    # local size = ceil(global/n); the last rank takes what is left (setup.cpp:478-495)
    sizes = [plan(3, r, (100, 64, 64), num_ranks=(3, 1, 1)) for r in range(3)]
    assert [s.local_size[0] for s in sizes] == [34, 34, 32]
    assert [s.rank_offset[0] for s in sizes] == [0, 34, 68]
    # local sizes given -> global derived
    p = plan(4, 3, local_size=(10, 20, 30), num_ranks=(2, 2, 1))
    assert tuple(p.global_size) == (20, 40, 30) and tuple(p.rank_offset) == (10, 20, 0)


def test_neighbor_sets_and_face_only_exchange_for_l1_1():
    p = plan(8, 0, (64, 64, 64))
    nb = neighbors(p)
    assert len(nb) == 7                       # a corner rank of a 2x2x2 grid
    faces = [n for n in nb if sum(abs(x) for x in n[1]) == 1]
    assert sorted(r for r, _ in faces) == [1, 2, 4]
    for r, o in nb:
        s = slab(p, o, l1=1)
        assert (s is not None) == (sum(abs(x) for x in o) == 1)   # iso3dfd p: faces only
        if s is not None:
            d = [i for i in range(3) if o[i]][0]
            assert s[1][d] == H and all(s[1][k] == 32 for k in range(3) if k != d)


# ------------------------------------------------------------------ (b) in-process emulation, 2x2x2, L1 = 2
def _exchange_all(plans, arrays, halo, l1):
    """Copy every planned send slab into the matching receive slab (what pack->transport->unpack does)."""
    n = len(plans)
    msgs = {}
    for r in range(n):
        for nr, o in neighbors(plans[r]):
            s = slab(plans[r], o, halo=(halo,) * 3, l1=l1, sending=True)
            if s is None:
                continue
            (f, sz) = s
            a = arrays[r]
            msgs[(r, nr)] = a[tuple(slice(halo + f[d], halo + f[d] + sz[d]) for d in range(3))].copy()
    for r in range(n):
        for nr, o in neighbors(plans[r]):
            s = slab(plans[r], o, halo=(halo,) * 3, l1=l1, sending=False)
            if s is None:
                assert (nr, r) not in msgs
                continue
            (f, sz) = s
            m = msgs[(nr, r)]
            assert m.shape == tuple(sz), "send slab of the neighbour and my receive slab disagree"
            arrays[r][tuple(slice(halo + f[d], halo + f[d] + sz[d]) for d in range(3))] = m


def test_diagonal_exchange_fills_every_in_domain_halo_cell():
    g, halo = (20, 24, 28), 3
    plans = [plan(8, r, g) for r in range(8)]
    arrays = []
    for p in plans:
        ls, ofs = tuple(p.local_size), tuple(p.rank_offset)
        full = O.fill(ls, halo, 5, 0, 0.0, 1.0, dtype=np.float32, origin=ofs)
        a = np.full_like(full, np.nan)
        a[halo:-halo, halo:-halo, halo:-halo] = full[halo:-halo, halo:-halo, halo:-halo]
        arrays.append((a, full))
    work = [a for a, _ in arrays]
    _exchange_all(plans, work, halo, l1=2)
    for (a, full), p in zip(arrays, plans):
        ls, ofs = tuple(p.local_size), tuple(p.rank_offset)
        # global coordinates of every cell of the halo'd box
        idx = np.meshgrid(*[np.arange(-halo, ls[d] + halo) + ofs[d] for d in range(3)], indexing="ij")
        inside = np.ones(a.shape, bool)
        for d in range(3):
            inside &= (idx[d] >= 0) & (idx[d] < g[d])
        # cells whose offset from the local box is non-zero in <= 2 dims are reachable with L1 = 2
        outside_dims = sum(((idx[d] < ofs[d]) | (idx[d] >= ofs[d] + ls[d])).astype(int) for d in range(3))
        need = inside & (outside_dims <= 2)
        assert not np.isnan(a[need]).any()
        assert np.array_equal(a[need], full[need])


def _emulate_iso3dfd(nranks, g, steps, num_ranks=(0, 0, 0)):
    plans = [plan(nranks, r, g, num_ranks=num_ranks) for r in range(nranks)]
    ids = O.VAR_IDS["iso3dfd"]
    init = O.DEFAULT_INIT["iso3dfd"]
    st = []
    for p in plans:
        ls, ofs = tuple(p.local_size), tuple(p.rank_offset)
        pp = [O.fill(ls, H, ids["p"], s, *init["p"], dtype=np.float32, origin=ofs) for s in (0, 1)]
        v = O.fill(ls, H, ids["v"], 0, *init["v"], dtype=np.float32, origin=ofs)
        st.append((pp, v, ls))
    fn = O.lib().yo_iso3dfd_step_f32
    for t in range(steps):
        for pp, v, ls in st:
            fn(O._ptr(pp[t % 2]), O._ptr(pp[(t + 1) % 2]), O._ptr(v), C.c_int64(ls[0]), C.c_int64(ls[1]), C.c_int64(ls[2]),
               C.c_int64(H), C.c_int(8))
        _exchange_all(plans, [pp[(t + 1) % 2] for pp, _, _ in st], H, l1=1)
    out = np.zeros(g, np.float32)
    for p, (pp, v, ls) in zip(plans, st):
        o = tuple(p.rank_offset)
        out[o[0]:o[0] + ls[0], o[1]:o[1] + ls[1], o[2]:o[2] + ls[2]] = O.interior(pp[steps % 2], H)
    return out


@pytest.mark.parametrize("nranks,num_ranks", [(8, (0, 0, 0)), (3, (3, 1, 1)), (4, (1, 2, 2))])
def test_emulated_rank_grid_equals_single_rank(nranks, num_ranks):
    g, steps = (40, 36, 44), 3
    ref = O.run_iso3dfd(g, steps)[("p", steps)]
    got = _emulate_iso3dfd(nranks, g, steps, num_ranks)
    assert np.array_equal(got, ref)          # identical arithmetic on identical inputs: bit-exact


# ------------------------------------------------------------------ (b2) the pipelined half-exchange schedule, emulated
def _emulate_iso3dfd_halves(nranks, g, steps, num_ranks, deliver):
    """Solution::run_stage_halves() in numpy: every rank steps its box in two launches -- the outer x-half [0, q1) u [q2, nx), then
    the inner half -- with the oracle's arithmetic; behind each goes the exchange of that half's part of the faces, cut by the
    library's own yk_plan_halves / yk_plan_halves_slab.  `deliver`: "late" = a half's messages land only after the NEXT half has
    been computed (what the library does: the next launch does not wait for them), "early" = they land before it (what a fast link
    does while that launch runs).  If both orders give the one-rank result bit for bit, the launch after a half-exchange neither
    needs nor disturbs it, and the launch after that finds everything it reads."""
    plans = [plan(nranks, r, g, num_ranks=num_ranks) for r in range(nranks)]
    ids, init = O.VAR_IDS["iso3dfd"], O.DEFAULT_INIT["iso3dfd"]
    st, cuts = [], []
    for p in plans:
        ls, ofs = tuple(p.local_size), tuple(p.rank_offset)
        pp = [O.fill(ls, H, ids["p"], s, *init["p"], dtype=np.float32, origin=ofs) for s in (0, 1)]
        v = O.fill(ls, H, ids["v"], 0, *init["v"], dtype=np.float32, origin=ofs)
        st.append((pp, v, ls))
        q1, q2 = _capi.idx_t(), _capi.idx_t()
        assert _lib().yk_plan_halves(ls[0], H, C.byref(q1), C.byref(q2)) == 1, "box too short in x for this test"
        cuts.append((q1.value, q2.value))
    fn = O.lib().yo_iso3dfd_step_f32

    def launch(r, t, half):
        (pp, v, ls), (q1, q2) = st[r], cuts[r]
        src, dst = pp[t % 2], pp[(t + 1) % 2]
        for a, b in ([(0, q1), (q2, ls[0])] if half == 0 else [(q1, q2)]):
            # the planes [a, b) with their x neighbours: a contiguous copy the oracle's whole-box step can work on
            s_sub, d_sub, v_sub = (np.ascontiguousarray(x[a:b + 2 * H]) for x in (src, dst, v))
            fn(O._ptr(s_sub), O._ptr(d_sub), O._ptr(v_sub), C.c_int64(b - a), C.c_int64(ls[1]), C.c_int64(ls[2]), C.c_int64(H), C.c_int(8))
            dst[a + H:b + H, H:-H, H:-H] = O.interior(d_sub, H)

    def pieces(r, o, sending, half):
        sl = slab(plans[r], o, sending=sending)
        if sl is None:
            return []
        (f, sz), (q1, q2) = sl, cuts[r]
        out4 = (_capi.idx_t * 4)()
        k = _lib().yk_plan_halves_slab(half, 1 if o[0] != 0 else 0, f[0], sz[0], q1, q2, out4)
        return [((out4[2 * i], f[1], f[2]), (out4[2 * i + 1], sz[1], sz[2])) for i in range(k)]

    def collect(t, half):        # what every rank sends behind this half: copies taken NOW (the pack kernel)
        msgs = {}
        for r in range(nranks):
            a = st[r][0][(t + 1) % 2]
            for nr, o in neighbors(plans[r]):
                msgs[(r, nr)] = [a[tuple(slice(H + f[d], H + f[d] + sz[d]) for d in range(3))].copy() for f, sz in pieces(r, o, True, half)]
        return (t, half, msgs)

    def land(flight):            # ... and where they land (the unpack kernel)
        t, half, msgs = flight
        for r in range(nranks):
            a = st[r][0][(t + 1) % 2]
            for nr, o in neighbors(plans[r]):
                got = msgs[(nr, r)]
                mine = pieces(r, o, False, half)
                assert len(got) == len(mine), "the two ends of a link cut the face differently"
                for m, (f, sz) in zip(got, mine):
                    assert m.shape == tuple(sz)
                    a[tuple(slice(H + f[d], H + f[d] + sz[d]) for d in range(3))] = m

    flight = None
    for t in range(steps):
        for half in (0, 1):
            if flight is not None and deliver == "early":
                land(flight); flight = None
            for r in range(nranks):
                launch(r, t, half)
            if flight is not None:
                land(flight)                     # finished only now: the launch above did not wait for it
            flight = collect(t, half)
    land(flight)
    out = np.zeros(g, np.float32)
    for p, (pp, v, ls) in zip(plans, st):
        o = tuple(p.rank_offset)
        out[o[0]:o[0] + ls[0], o[1]:o[1] + ls[1], o[2]:o[2] + ls[2]] = O.interior(pp[steps % 2], H)
    return out


def _emulate_ssg_halves(nranks, g, steps, num_ranks, deliver):
    """The same pipeline for a two-stage solution with in-place fields (ssg: stage 1 updates three velocities from six stresses, stage
    2 the six stresses from the NEW velocities): half-launches S1-A, S1-B, S2-A, S2-B, S1-A ...; behind each, the exchange of what its
    stage wrote, cut at the halves' planes.  (Model: every field travels 4 wide on all faces -- the library sends each field's own,
    narrower halos; the coefficient fields are constant and filled with their halos.)"""
    HS = 4
    plans = [plan(nranks, r, g, num_ranks=num_ranks) for r in range(nranks)]
    ids, init = O.VAR_IDS["ssg"], O.DEFAULT_INIT["ssg"]
    st, cuts = [], []
    for p in plans:
        ls, ofs = tuple(p.local_size), tuple(p.rank_offset)
        f = {n: O.fill(ls, HS, ids[n], 0, *init[n], dtype=np.float32, origin=ofs) for n in O.SSG_FIELDS}
        k = {n: O.fill(ls, HS, ids[n], 0, *init[n], dtype=np.float32, origin=ofs) for n in O.SSG_COEFFS}
        st.append((f, k, ls))
        q1, q2 = _capi.idx_t(), _capi.idx_t()
        assert _lib().yk_plan_halves(ls[0], HS, C.byref(q1), C.byref(q2)) == 1
        cuts.append((q1.value, q2.value))
    stage_fn = [O.lib().yo_ssg_stage1_f32, O.lib().yo_ssg_stage2_f32]
    written = [O.SSG_FIELDS[:3], O.SSG_FIELDS[3:]]

    def launch(r, stage, half):
        (f, k, ls), (q1, q2) = st[r], cuts[r]
        for a, b in ([(0, q1), (q2, ls[0])] if half == 0 else [(q1, q2)]):
            fs = {n: np.ascontiguousarray(f[n][a:b + 2 * HS]) for n in O.SSG_FIELDS}
            ks = {n: np.ascontiguousarray(k[n][a:b + 2 * HS]) for n in O.SSG_COEFFS}
            fa = (C.c_void_p * 9)(*[fs[n].ctypes.data for n in O.SSG_FIELDS])
            ka = (C.c_void_p * 4)(*[ks[n].ctypes.data for n in O.SSG_COEFFS])
            stage_fn[stage](fa, ka, C.c_int64(b - a), C.c_int64(ls[1]), C.c_int64(ls[2]), C.c_int64(HS))
            for n in written[stage]:
                f[n][a + HS:b + HS, HS:-HS, HS:-HS] = O.interior(fs[n], HS)

    def pieces(r, o, sending, half):
        sl = slab(plans[r], o, halo=(HS, HS, HS), sending=sending)
        if sl is None:
            return []
        (f, sz), (q1, q2) = sl, cuts[r]
        out4 = (_capi.idx_t * 4)()
        kk = _lib().yk_plan_halves_slab(half, 1 if o[0] != 0 else 0, f[0], sz[0], q1, q2, out4)
        return [((out4[2 * i], f[1], f[2]), (out4[2 * i + 1], sz[1], sz[2])) for i in range(kk)]

    def collect(stage, half):
        msgs = {}
        for r in range(nranks):
            for nr, o in neighbors(plans[r]):
                msgs[(r, nr)] = [[st[r][0][n][tuple(slice(HS + f[d], HS + f[d] + sz[d]) for d in range(3))].copy() for n in written[stage]]
                                 for f, sz in pieces(r, o, True, half)]
        return (stage, half, msgs)

    def land(flight):
        stage, half, msgs = flight
        for r in range(nranks):
            for nr, o in neighbors(plans[r]):
                got, mine = msgs[(nr, r)], pieces(r, o, False, half)
                assert len(got) == len(mine)
                for per_field, (f, sz) in zip(got, mine):
                    for n, m in zip(written[stage], per_field):
                        assert m.shape == tuple(sz)
                        st[r][0][n][tuple(slice(HS + f[d], HS + f[d] + sz[d]) for d in range(3))] = m

    flight = None
    for _ in range(steps):
        for stage in (0, 1):
            for half in (0, 1):
                if flight is not None and deliver == "early":
                    land(flight); flight = None
                for r in range(nranks):
                    launch(r, stage, half)
                if flight is not None:
                    land(flight)
                flight = collect(stage, half)
    land(flight)
    out = {n: np.zeros(g, np.float32) for n in O.SSG_FIELDS}
    for p, (f, k, ls) in zip(plans, st):
        o = tuple(p.rank_offset)
        for n in O.SSG_FIELDS:
            out[n][o[0]:o[0] + ls[0], o[1]:o[1] + ls[1], o[2]:o[2] + ls[2]] = O.interior(f[n], HS)
    return out


@pytest.mark.parametrize("deliver", ["late", "early"])
@pytest.mark.parametrize("nranks,num_ranks,g", [(8, (2, 2, 2), (40, 28, 36)), (4, (1, 2, 2), (24, 28, 36))])
def test_emulated_pipelined_half_exchanges_two_stages_in_place(nranks, num_ranks, g, deliver):
    steps = 2
    ref = O.run_ssg(g, steps)
    got = _emulate_ssg_halves(nranks, g, steps, num_ranks, deliver)
    for n in O.SSG_FIELDS:
        assert np.array_equal(got[n], ref[(n, steps)]), n


@pytest.mark.parametrize("deliver", ["late", "early"])
@pytest.mark.parametrize("nranks,num_ranks,g", [(8, (2, 2, 2), (64, 36, 44)), (4, (1, 2, 2), (40, 36, 44)), (4, (2, 2, 1), (70, 36, 30))])
def test_emulated_pipelined_half_exchanges_equal_single_rank(nranks, num_ranks, g, deliver):
    steps = 3
    ref = O.run_iso3dfd(g, steps)[("p", steps)]
    got = _emulate_iso3dfd_halves(nranks, g, steps, num_ranks, deliver)
    assert np.array_equal(got, ref)


# ------------------------------------------------------------------ (c) real processes, gloo, world size 2
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, g, steps, num_ranks, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = plan(world, rank, g, num_ranks=num_ranks)
        ls, ofs = tuple(p.local_size), tuple(p.rank_offset)
        ids, init = O.VAR_IDS["iso3dfd"], O.DEFAULT_INIT["iso3dfd"]
        pp = [O.fill(ls, H, ids["p"], s, *init["p"], dtype=np.float32, origin=ofs) for s in (0, 1)]
        v = O.fill(ls, H, ids["v"], 0, *init["v"], dtype=np.float32, origin=ofs)
        fn = O.lib().yo_iso3dfd_step_f32
        for t in range(steps):
            fn(O._ptr(pp[t % 2]), O._ptr(pp[(t + 1) % 2]), O._ptr(v), C.c_int64(ls[0]), C.c_int64(ls[1]), C.c_int64(ls[2]),
               C.c_int64(H), C.c_int(8))
            a = pp[(t + 1) % 2]
            ops, recvs = [], []
            for nr, o in neighbors(p):
                s = slab(p, o, sending=True)
                r = slab(p, o, sending=False)
                if s is not None:
                    (f, sz) = s
                    buf = torch.from_numpy(np.ascontiguousarray(a[tuple(slice(H + f[d], H + f[d] + sz[d]) for d in range(3))]))
                    ops.append(dist.P2POp(dist.isend, buf, nr))
                if r is not None:
                    (f, sz) = r
                    rb = torch.empty(tuple(sz), dtype=torch.float32)
                    ops.append(dist.P2POp(dist.irecv, rb, nr))
                    recvs.append((f, sz, rb))
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            for f, sz, rb in recvs:
                a[tuple(slice(H + f[d], H + f[d] + sz[d]) for d in range(3))] = rb.numpy()
        q.put((rank, ofs, ls, O.interior(pp[steps % 2], H).copy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("num_ranks", [(2, 1, 1), (1, 1, 2)])
def test_two_process_gloo_halo_exchange_equals_single_rank(num_ranks):
    import torch.multiprocessing as mp
    g, steps = (34, 30, 40), 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, g, steps, num_ranks, q)) for r in range(2)]
    for p in procs:
        p.start()
    parts = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = O.run_iso3dfd(g, steps)[("p", steps)]
    got = np.zeros(g, np.float32)
    for _, o, ls, a in parts:
        got[o[0]:o[0] + ls[0], o[1]:o[1] + ls[1], o[2]:o[2] + ls[2]] = a
    assert np.array_equal(got, ref)


def _worker_halves(rank, world, port, g, steps, num_ranks, q):
    """the pipelined half-exchange schedule between REAL processes: a half's messages are posted (isend / irecv of the cut faces)
    behind its launch and completed only after the next half has been computed"""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = plan(world, rank, g, num_ranks=num_ranks)
        ls, ofs = tuple(p.local_size), tuple(p.rank_offset)
        ids, init = O.VAR_IDS["iso3dfd"], O.DEFAULT_INIT["iso3dfd"]
        pp = [O.fill(ls, H, ids["p"], s, *init["p"], dtype=np.float32, origin=ofs) for s in (0, 1)]
        v = O.fill(ls, H, ids["v"], 0, *init["v"], dtype=np.float32, origin=ofs)
        fn = O.lib().yo_iso3dfd_step_f32
        q1, q2 = _capi.idx_t(), _capi.idx_t()
        assert _lib().yk_plan_halves(ls[0], H, C.byref(q1), C.byref(q2)) == 1
        q1, q2 = q1.value, q2.value

        def launch(t, half):
            src, dst = pp[t % 2], pp[(t + 1) % 2]
            for a, b in ([(0, q1), (q2, ls[0])] if half == 0 else [(q1, q2)]):
                s_sub, d_sub, v_sub = (np.ascontiguousarray(x[a:b + 2 * H]) for x in (src, dst, v))
                fn(O._ptr(s_sub), O._ptr(d_sub), O._ptr(v_sub), C.c_int64(b - a), C.c_int64(ls[1]), C.c_int64(ls[2]), C.c_int64(H), C.c_int(8))
                dst[a + H:b + H, H:-H, H:-H] = O.interior(d_sub, H)

        def pieces(o, sending, half):
            sl = slab(p, o, sending=sending)
            if sl is None:
                return []
            f, sz = sl
            out4 = (_capi.idx_t * 4)()
            k = _lib().yk_plan_halves_slab(half, 1 if o[0] != 0 else 0, f[0], sz[0], q1, q2, out4)
            return [((out4[2 * i], f[1], f[2]), (out4[2 * i + 1], sz[1], sz[2])) for i in range(k)]

        def start(t, half):
            a = pp[(t + 1) % 2]
            ops, recvs, keep = [], [], []
            for nr, o in neighbors(p):
                for f, sz in pieces(o, True, half):
                    buf = torch.from_numpy(np.ascontiguousarray(a[tuple(slice(H + f[d], H + f[d] + sz[d]) for d in range(3))]))
                    keep.append(buf)
                    ops.append(dist.P2POp(dist.isend, buf, nr))
                for f, sz in pieces(o, False, half):
                    rb = torch.empty(tuple(sz), dtype=torch.float32)
                    ops.append(dist.P2POp(dist.irecv, rb, nr))
                    recvs.append((f, sz, rb))
            return (dist.batch_isend_irecv(ops) if ops else [], recvs, a, keep)

        def finish(flight):
            reqs, recvs, a, _ = flight
            for w in reqs:
                w.wait()
            for f, sz, rb in recvs:
                a[tuple(slice(H + f[d], H + f[d] + sz[d]) for d in range(3))] = rb.numpy()

        flight = None
        for t in range(steps):
            for half in (0, 1):
                launch(t, half)
                if flight is not None:
                    finish(flight)             # the previous half's exchange: the launch above did not wait for it
                flight = start(t, half)
        finish(flight)
        q.put((rank, ofs, ls, O.interior(pp[steps % 2], H).copy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("num_ranks,g", [((1, 1, 2), (40, 30, 40)), ((1, 2, 1), (36, 40, 30))])
def test_two_process_gloo_pipelined_half_exchanges_equal_single_rank(num_ranks, g):
    import torch.multiprocessing as mp
    steps = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_halves, args=(r, 2, port, g, steps, num_ranks, q)) for r in range(2)]
    for p in procs:
        p.start()
    parts = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = O.run_iso3dfd(g, steps)[("p", steps)]
    got = np.zeros(g, np.float32)
    for _, o, ls, a in parts:
        got[o[0]:o[0] + ls[0], o[1]:o[1] + ls[1], o[2]:o[2] + ls[2]] = a
    assert np.array_equal(got, ref)


# ------------------------------------------------------------------ (d) native rank bootstrap: the TCP rendezvous
def _rdv_worker(rank, world, port, q):
    lib = _capi.load("iso3dfd")        # dlopen only
    buf = C.create_string_buffer(bytes(range(128)) if rank == 0 else bytes(128), 128)
    rc = lib.yk_rendezvous_bcast(rank, world, b"127.0.0.1", port, buf, 128)
    q.put((rank, rc, buf.raw))


def test_native_rendezvous_distributes_the_unique_id():
    """yk_env_init_from_launcher() (the C++ new_env() path, yask_amd/csrc/ykh_launch.cpp) hands the 128-byte
    ncclUniqueId from rank 0 to the other ranks over a one-shot TCP rendezvous -- no MPI, no torch.  The rendezvous
    itself needs no GPU: three processes, ranks 1 and 2 start before rank 0 listens."""
    import multiprocessing as mp
    import time
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_rdv_worker, args=(r, 3, port, q)) for r in (1, 2)]
    for p in procs:
        p.start()
    time.sleep(0.5)                     # the clients retry until the server is up
    p0 = ctx.Process(target=_rdv_worker, args=(0, 3, port, q))
    p0.start()
    got = dict((r, (rc, raw)) for r, rc, raw in (q.get(timeout=60) for _ in range(3)))
    for p in procs + [p0]:
        p.join(timeout=30)
        assert p.exitcode == 0
    for r in range(3):
        assert got[r] == (0, bytes(range(128))), r
    # world size 1 is a no-op
    assert _capi.load("iso3dfd").yk_rendezvous_bcast(0, 1, b"127.0.0.1", port, C.create_string_buffer(8), 8) == 0


def _mesh_worker(rank, world, port, q):
    lib = _capi.load("iso3dfd")        # dlopen only
    total = C.c_longlong(-1)
    rc = lib.yk_tcp_mesh_check(rank, world, b"127.0.0.1", port, C.byref(total))
    q.put((rank, rc, total.value))


def test_tcp_mesh_of_eight_ranks_connects_every_time():
    """The full mesh of the host-staged TCP transport (what lets 2 ... 8 ranks share the one GPU of a test box): every rank
    listens on a kernel-assigned port and rank 0 hands out the table.  (Fixed listener ports base + rank collided now and then
    with the ephemeral source ports of the other ranks' outgoing connections: one 8-rank GPU test in ~30 died with "could not
    connect the mesh".)  Eight processes, several rounds in a row, all ranks started at once; device-free."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    world = 8
    for rnd in range(6):
        q = ctx.Queue()
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        procs = [ctx.Process(target=_mesh_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        got = sorted(q.get(timeout=120) for _ in range(world))
        for p in procs:
            p.join(timeout=30)
            assert p.exitcode == 0
        assert got == [(r, 0, world * (world - 1) // 2) for r in range(world)], (rnd, got)


# ------------------------------------------------------------------ (e) wave-front temporal tiling: the launch plan
def _wavefront(lo, hi, width, angle, nphases):
    n = _lib().yk_plan_wavefront(lo, hi, width, angle, nphases, None, 0)
    buf = (_capi.idx_t * (3 * max(1, n)))()
    assert _lib().yk_plan_wavefront(lo, hi, width, angle, nphases, buf, n) == n
    return [(buf[3 * i], buf[3 * i + 1], buf[3 * i + 2]) for i in range(n)]


@pytest.mark.parametrize("nx,width,radius,nsteps,nstages", [(97, 16, 4, 3, 1), (64, 8, 8, 2, 1), (50, 24, 3, 4, 2), (30, 64, 2, 5, 1)])
def test_wavefront_plan_reproduces_plain_sweeps_in_place(nx, width, radius, nsteps, nstages):
    """Solution::run_wavefront() launches (phase, [lo, hi)) boxes in the order yk_plan_wavefront() gives.  Emulated in 1-D
    with numpy and the SAME in-place storage rules as the library -- a 2-slot var whose step t+1 overwrites step t-1
    (iso3dfd-like) or two 1-slot vars updated in place by two stages (ssg-like): the schedule must give exactly what
    plain sweeps give, and every phase must cover [0, nx) exactly once."""
    rng = np.random.default_rng(1)
    r = radius
    launches = _wavefront(0, nx, width, r, nsteps * nstages)
    for p in range(nsteps * nstages):          # exact cover per phase
        cover = np.zeros(nx, int)
        for ph, a, b in launches:
            if ph == p:
                cover[a:b] += 1
        assert (cover == 1).all(), p
    w = rng.standard_normal(2 * r + 1) * 0.2
    if nstages == 1:
        def run(schedule):
            u = [rng0.copy() for rng0 in init]          # two slots, pads of width r never written
            for ph, a, b in schedule:
                t = ph
                src, dst = u[t % 2], u[(t + 1) % 2]
                new = np.array([2 * src[r + x] - dst[r + x] + (w * src[x:x + 2 * r + 1]).sum() for x in range(a, b)])
                dst[r + a:r + b] = new
            return u
        init = [rng.standard_normal(nx + 2 * r), rng.standard_normal(nx + 2 * r)]
        plain = run([(p, 0, nx) for p in range(nsteps)])
        wf = run(launches)
    else:
        def run(schedule):
            f, g = init[0].copy(), init[1].copy()       # stage 1: f += W*g ; stage 2: g += W*f (new f), both in place
            for ph, a, b in schedule:
                src, dst = (g, f) if ph % 2 == 0 else (f, g)
                new = np.array([dst[r + x] + (w * src[x:x + 2 * r + 1]).sum() for x in range(a, b)])
                dst[r + a:r + b] = new
            return [f, g]
        init = [rng.standard_normal(nx + 2 * r), rng.standard_normal(nx + 2 * r)]
        plain = run([(p, 0, nx) for p in range(nsteps * nstages)])
        wf = run(launches)
    for a, b in zip(plain, wf):
        assert np.array_equal(a, b)


# ------------------------------------------------------------------ (f) planned launch of a decomposed rank: the block list
def _plan(n, lo, hi, width=(8, 8, 8), ty=32, tz=128, overhead=9, ncu=256, shell_pct=55, mode=0):
    A3, I3 = _capi.idx_t * 3, C.c_int * 3
    info = (_capi.idx_t * 5)()
    args = (A3(*n), I3(*lo), I3(*hi), A3(*width), ty, tz, overhead, ncu, shell_pct, mode)
    cnt = _lib().yk_plan_blocks(*args, None, 0, info)
    assert cnt > 0
    buf = (_capi.BlockDesc * cnt)()
    assert _lib().yk_plan_blocks(*args, buf, cnt, info) == cnt
    return [(b.x0, b.x1, b.y0, b.y1, b.z0, b.z1, b.flags, b.start) for b in buf], list(info)


@pytest.mark.parametrize("n,lo,hi,ty,tz", [((40, 44, 72), (0, 0, 0), (1, 1, 1), 16, 32), ((64, 30, 50), (1, 1, 1), (1, 1, 1), 8, 16),
                                           ((33, 17, 129), (1, 0, 0), (0, 0, 1), 32, 128), ((48, 48, 48), (0, 0, 0), (0, 0, 0), 16, 16),
                                           ((24, 64, 64), (0, 1, 0), (0, 0, 0), 16, 64), ((20, 10, 200), (1, 0, 1), (1, 0, 0), 4, 64)])
def test_block_plan_covers_the_rank_box_exactly_once_shell_first(n, lo, hi, ty, tz, mode=0):
    """Solution::launch_planned() hands workgroup i the i-th descriptor of yk_plan_blocks().  Whatever the plan, every point of
    the rank box must be computed exactly once, every block must be ONE tile of the regular (ty, tz) tiling (the kernel's
    threads cover exactly that), the shell -- every point a neighbour needs: within `width` of a face that has one -- must be
    made of signalling blocks only, and those come first in dispatch order."""
    w = (3, 4, 5)
    blocks, info = _plan(n, lo, hi, width=w, ty=ty, tz=tz, overhead=5, ncu=24, mode=mode)
    count = np.zeros(n, np.int32)
    sig = np.zeros(n, bool)
    seen_plain = False
    for x0, x1, y0, y1, z0, z1, flags, start in blocks:
        assert 0 <= x0 < x1 <= n[0] and 0 <= y0 < y1 <= n[1] and 0 <= z0 < z1 <= n[2]
        assert y0 % ty == 0 and y1 <= y0 + ty and z0 % tz == 0 and z1 <= z0 + tz          # one tile of the regular tiling
        assert y1 == min(y0 + ty, n[1]) and z1 == min(z0 + tz, n[2])                         # ... all of it
        count[x0:x1, y0:y1, z0:z1] += 1
        if flags & 1:
            assert not seen_plain, "a signalling block after the interior has begun"
            sig[x0:x1, y0:y1, z0:z1] = True
        else:
            seen_plain = True
    assert (count == 1).all()
    assert sum(1 for b in blocks if b[6] & 1) == info[0]
    need = np.zeros(n, bool)
    for d in range(3):
        idx = [slice(None)] * 3
        if lo[d]:
            idx[d] = slice(0, w[d]); need[tuple(idx)] = True
        if hi[d]:
            idx[d] = slice(n[d] - w[d], n[d]); need[tuple(idx)] = True
    assert (sig | ~need).all(), "a point a neighbour needs is computed by a block that does not signal"
    if not any(lo) and not any(hi):
        assert info[0] == 0
    assert info[2] >= info[1] >= 0 and info[3] > 0


def test_block_plans_of_the_baseline_blocks_stay_close_to_the_undivided_sweep():
    """The cost model's verdict for the blocks BASELINE configs 2 / 4 give a GPU of the 2x2x2 grid (iso3dfd: tile 128 x 32,
    9 plane-iterations of prologue per block, 256 CUs): simulated makespan of the planned launch vs the same box as one regular
    launch.  Round 2's separate launches measured 1.22-1.51x (512^3) and 1.18-1.29x (1024 x 1024 x 512) on the GPU."""
    for n, bound in (((512, 512, 512), 1.08), ((1024, 1024, 512), 1.03), ((512, 1024, 1024), 1.03)):
        hi = (1, 1, 1) if n[1] == n[0] or n[2] == 512 else (1, 0, 0)
        blocks, info = _plan(n, (0, 0, 0), hi)
        n_sig, shell_done, makespan, undivided, mode = info
        print(n, "blocks", len(blocks), "shell blocks", n_sig, "shell done", shell_done, "end", makespan, "undivided", undivided, "mode", mode)
        assert makespan <= bound * undivided, (n, makespan, undivided)
        assert shell_done <= 0.75 * makespan          # the exchange gets at least a quarter of the launch to hide in


# ------------------------------------------------------------------ (g) pipelined half-exchanges: the two launches and the cut faces
@pytest.mark.parametrize("n,ty,tz,ncu", [((64, 44, 72), 16, 32, 24), ((512, 512, 512), 32, 128, 256), ((1024, 1024, 512), 32, 128, 256),
                                         ((130, 30, 50), 8, 16, 7), ((512, 512, 512), 16, 128, 256), ((37, 17, 129), 32, 128, 256)])
def test_halves_plan_is_the_regular_tiling_in_two_launches(n, ty, tz, ncu):
    """Plan mode 4 (Solution::run_stage_halves, -hip_halves): every point of the rank box exactly once, whole tiles of the regular
    tiling, the first `cut` blocks cover exactly the outer x-half [0, q1) u [q2, nx), the rest the inner half; no block signals
    (nothing is special: the exchange of a half starts when its launch has ended); q1, q2 come from the x extent alone."""
    q1, q2 = _capi.idx_t(), _capi.idx_t()
    assert _lib().yk_plan_halves(n[0], 8, C.byref(q1), C.byref(q2)) == 1
    q1, q2 = q1.value, q2.value
    assert 8 <= q1 == n[0] // 4 and q2 == n[0] - q1
    blocks, info = _plan(n, (0, 1, 0), (1, 1, 1), ty=ty, tz=tz, overhead=9, ncu=ncu, mode=4)
    cut, mode = info[0], info[4]
    assert mode == 4 and 0 < cut < len(blocks)
    count = np.zeros(n, np.int32)
    for i, (x0, x1, y0, y1, z0, z1, flags, start) in enumerate(blocks):
        assert 0 <= x0 < x1 <= n[0] and flags == 0
        assert y0 % ty == 0 and z0 % tz == 0 and y1 == min(y0 + ty, n[1]) and z1 == min(z0 + tz, n[2])
        outer = x1 <= q1 or x0 >= q2
        inner = x0 >= q1 and x1 <= q2
        assert (outer if i < cut else inner), (i, cut, x0, x1, q1, q2)
        count[x0:x1, y0:y1, z0:z1] += 1
    assert (count == 1).all()
    # all blocks of a launch have (nearly) the same length: they start and end together
    for part in (blocks[:cut], blocks[cut:]):
        lens = [b[1] - b[0] for b in part]
        assert max(lens) - min(lens) <= 1


def test_halves_plans_of_the_baseline_blocks_cost_what_the_undivided_sweep_costs():
    """Cost model: the two launches of the halves schedule against the same box as one regular launch (plane-iterations, 256 CUs,
    iso3dfd's tile and 9 plane-iterations of prologue).  The shell-first plans of mode 0 measured 1.05-1.13x on the GPU; the same
    blocks in regular order 1.01-1.02x -- which is what these launches are."""
    for n, ty in (((512, 512, 512), 32), ((1024, 1024, 512), 32), ((512, 512, 1024), 32), ((512, 512, 512), 16)):
        blocks, info = _plan(n, (0, 0, 0), (1, 1, 1), ty=ty, mode=4)
        cut, half_a, makespan, undivided, mode = info
        print(n, "tile y", ty, "blocks", len(blocks), "first launch", cut, "ends at", half_a, "both", makespan, "undivided", undivided)
        assert makespan <= 1.08 * undivided
        assert 0.4 * makespan <= half_a <= 0.6 * makespan       # two halves: each transfer has the other launch to hide behind
        assert cut % 256 == 0 and (len(blocks) - cut) % 256 == 0  # whole rounds of workgroups in both launches


def test_halves_cut_a_face_at_the_same_planes_on_both_sides_and_x_faces_travel_first():
    def ranges(half, xnb, lo, n, q1, q2):
        out = (_capi.idx_t * 4)()
        k = _lib().yk_plan_halves_slab(half, xnb, lo, n, q1, q2, out)
        return [(out[2 * i], out[2 * i + 1]) for i in range(k)]
    nx, q1, q2 = 512, 128, 384
    # a y / z face: whole x extent; the two halves partition it
    assert ranges(0, 0, 0, nx, q1, q2) == [(0, 128), (384, 128)]
    assert ranges(1, 0, 0, nx, q1, q2) == [(128, 256)]
    # ... also when the slab reaches into the rank's own x halo at a global boundary (vars read diagonally: ssg's mu)
    assert ranges(0, 0, -1, nx + 2, q1, q2) == [(-1, 129), (384, 129)]
    assert ranges(1, 0, -1, nx + 2, q1, q2) == [(128, 256)]
    # x faces (and edges / corners that are offset in x) travel whole with the outer half
    assert ranges(0, 1, 0, 8, q1, q2) == [(0, 8)] and ranges(1, 1, 0, 8, q1, q2) == []
    assert ranges(0, 1, nx - 8, 8, q1, q2) == [(nx - 8, 8)] and ranges(1, 1, nx - 8, 8, q1, q2) == []
    # a var without the x dim (extent 1 at x = 0): once, with the outer half
    assert ranges(0, 0, 0, 1, q1, q2) == [(0, 1)] and ranges(1, 0, 0, 1, q1, q2) == []
    # every plane of a slab is in exactly one range of exactly one half
    rng = np.random.default_rng(5)
    for _ in range(200):
        nx = int(rng.integers(32, 700))
        a, b = _capi.idx_t(), _capi.idx_t()
        assert _lib().yk_plan_halves(nx, 8, C.byref(a), C.byref(b)) == 1
        lo = int(rng.integers(-4, 1)); n = nx - lo + int(rng.integers(0, 5))
        seen = np.zeros(n, np.int32)
        for h in (0, 1):
            for l, m in ranges(h, 0, lo, n, a.value, b.value):
                assert m > 0
                seen[l - lo:l - lo + m] += 1
        assert (seen == 1).all()
    # too short in x for an outer quarter that holds the x halo: no halves
    a, b = _capi.idx_t(), _capi.idx_t()
    assert _lib().yk_plan_halves(24, 8, C.byref(a), C.byref(b)) == 0
    assert _lib().yk_plan_halves(32, 8, C.byref(a), C.byref(b)) == 1 and (a.value, b.value) == (8, 24)


def test_block_plan_refuses_the_deleted_planner_modes():
    """round 5: modes 1 / 2 (thin x slabs with per-CU budgets / uniform interior chunks, measured 1.3-1.6x) are gone"""
    A3, I3 = _capi.idx_t * 3, C.c_int * 3
    for mode in (1, 2, 7):
        info = (_capi.idx_t * 5)()
        assert _lib().yk_plan_blocks(A3(40, 44, 72), I3(0, 0, 0), I3(1, 1, 1), A3(8, 8, 8), 32, 128, 9, 256, 55, mode, None, 0, info) < 0
        assert b"mode must be 0" in _lib().yk_last_error()
        _lib().yk_clear_error()
