"""2-D solutions cut over ranks (round 6).

The reference runs every stencil of its matrix at `ranks > 1` as well (src/kernel/Makefile:1042-1062); until round 6 no test here cut
a 2-D solution over ranks at all, and since round 6 their parts run on lifted vector kernels (csrc/ykh_lift2d.hpp), get bounding
boxes and ring strips for their conditional scratch parts, and may be fused (one rank only: a decomposed run must fall back to the
part-by-part path by itself).  N processes on one device (IPC transport, launcher bootstrap); the checker is the reference's own
one-rank result (tests/golden/*_40x520_*), every rank's box of every written var ≤ 2e-5 of max|ref|."""
import json
import os
import socket
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden"
INDEX = json.load(open(G / "index.json"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, name, nr, q, fuse=None):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      YASK_HIP_TRANSPORT="ipc", YASK_HIP_WAIT_TIMEOUT_S="30", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if fuse is None:
        os.environ.pop("YASK_HIP_FUSE_SCRATCH", None)
    else:
        os.environ["YASK_HIP_FUSE_SCRATCH"] = str(fuse)
    from yask_amd import yk_factory
    meta = INDEX[name]
    fac = yk_factory(meta["stencil"])
    env = fac.new_env()
    env.init_from_launcher()
    s = fac.new_solution(env)
    s.set_overall_domain_size_vec(meta["size"])
    s.set_num_ranks_vec(list(nr))
    s.prepare_solution()
    if fuse is not None:
        assert len(s.get_fused_groups()) == (1 if fuse else 0), (fuse, s.get_fused_groups())
    for i, v in enumerate(s.get_vars()):
        off, sc = meta.get("init_vars", {}).get(v.get_name(), meta["init"])
        v.set_elements_hash(off, sc, hash_id=i)
    s.run_solution(0, meta["steps"] - 1)
    f, l = s.get_first_rank_domain_index_vec(), s.get_last_rank_domain_index_vec()
    out = {}
    for key in meta["arrays"]:
        vname, t = key.split("@")
        var = s.get_var(vname)
        dn = var.get_dim_names()
        if len(dn) != len(meta["size"]) + 1 or dn[0] != s.get_step_dim_name():
            continue                                     # (coefficient scalars: inputs)
        out[key] = np.asarray(var.get_elements_in_slice([int(t)] + f, [int(t)] + l))[0]
    q.put((rank, f, l, out, [s.get_kernel_variant(p) for p in range(s.get_num_parts())]))
    env.global_barrier()
    s.end_solution()


@pytest.mark.parametrize("stencil,world,nr", [(st, w, nr) for st in ("wave2d", "swe2d", "test_boundary_2d", "test_scratch_2d")
                                              for w, nr in ((2, (2, 1)), (2, (1, 2)), (4, (2, 2)))] +
                         [(st, w, (w,)) for st in ("test_1d", "test_boundary_1d", "test_scratch_boundary_1d") for w in (2, 4)],
                         ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else str(v))
def test_two_d_solution_over_ranks_matches_the_reference(gpu, stencil, world, nr):
    _ranks_vs_fixture(stencil, world, nr, None)


@pytest.mark.parametrize("fuse", [1, 0], ids=["fused", "part_by_part"])
@pytest.mark.parametrize("stencil,world,nr", [("wave2d", 2, (1, 2)), ("swe2d", 4, (2, 2)), ("test_scratch_2d", 2, (2, 1))], ids=["wave2d-1x2", "swe2d-2x2", "test_scratch_2d-2x1"])
def test_fused_scratch_groups_on_decomposed_ranks(gpu, stencil, world, nr, fuse):
    """a decomposed rank runs a fused group over its whole box and exchanges afterwards (the scratch vars never travel: every rank computes
    them on its box grown by their halos): forced on and forced off, both against the reference's one-rank result"""
    _ranks_vs_fixture(stencil, world, nr, fuse)


def _ranks_vs_fixture(stencil, world, nr, fuse):
    """(and, since the lifted vector kernels serve them too, three of the 1-D solutions over 2 and 4 ranks on their 2300-point goldens)"""
    import multiprocessing as mp
    want = [40, 520] if len(nr) == 2 else [2300]
    name = [n for n in INDEX if INDEX[n].get("generic") and INDEX[n]["stencil"] == stencil and INDEX[n]["size"] == want][0]
    meta = INDEX[name]
    z = np.load(G / f"{name}.npz")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, nr, q, fuse)) for r in range(world)]
    for p in procs:
        p.start()
    parts = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    checked = 0
    for key in meta["arrays"]:
        ref = z[key].astype(np.float64)
        if ref.ndim != len(nr):
            continue
        got = np.full(ref.shape, np.nan)
        have = False
        for _, f, l, out, _ in parts:
            if key in out:
                got[tuple(slice(f0, l0 + 1) for f0, l0 in zip(f, l))] = out[key]
                have = True
        if not have:
            continue
        assert np.isfinite(got).all(), (key, "the ranks' boxes cover the domain")
        err = np.abs(got - ref).max() / max(1e-30, np.abs(ref).max())
        assert err <= 2e-5, (key, err)
        checked += 1
    assert checked >= 1
