"""N>1 data path on the GPU: two ranks (two processes) share the one GPU of the test box and run the REAL
library path -- exterior/interior split, pack kernels, halo transport, unpack kernels, dirty-flag
bookkeeping, comm/compute stream ordering -- with the host-staged torch.distributed/gloo transport
(RCCL needs one device per rank, so its send/recv calls themselves are exercised only by the driver's
multi-GPU bench).  The union of the rank domains must equal the single-rank GPU result BIT-exactly (same
kernel, same per-point arithmetic) and the oracle within the stated tolerance."""
import os
import socket

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _fields(stencil):
    return {"iso3dfd": ["p"], "3axis": ["A"], "ssg": O.SSG_FIELDS}[stencil]


def _run_rank(stencil, g, steps, opts, nr):
    """Create, init and run a solution in this process; returns (soln, local first index, local sizes)."""
    from yask_amd import yk_factory, dist as ydist
    fac = yk_factory(stencil)
    env, _ = ydist.new_env(fac, "torch")
    soln = fac.new_solution(env)
    soln.set_overall_domain_size_vec(list(g))
    if nr is not None:
        soln.set_num_ranks_vec(list(nr))
    if opts:
        assert soln.apply_command_line_options(opts) == ""
    soln.prepare_solution()
    init = O.DEFAULT_INIT[stencil]
    for v in soln.get_vars():
        v.set_elements_hash(*init[v.get_name()], hash_id=O.VAR_IDS[stencil][v.get_name()])
    soln.run_solution(0, steps - 1)
    return soln


def _local_result(soln, stencil, steps):
    f = soln.get_first_rank_domain_index_vec()
    l = soln.get_last_rank_domain_index_vec()
    out = {}
    for n in _fields(stencil):
        v = soln.get_var(n)
        a = v.get_elements_in_slice([steps] + f, [steps] + l)[0]
        out[n] = a
    return f, out


def _worker(rank, world, port, stencil, g, steps, opts, nr, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from yask_amd import dist as ydist
    torch.cuda.set_device(0)
    ydist.init_process_group(backend="gloo")
    try:
        soln = _run_rank(stencil, g, steps, opts, nr)
        assert soln.get_num_ranks_vec() == list(nr)
        f, out = _local_result(soln, stencil, steps)
        q.put((rank, f, out))
        dist.barrier()
        soln.end_solution()
    finally:
        dist.destroy_process_group()


def _two_ranks(stencil, g, steps, opts, nr):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, stencil, g, steps, opts, nr, q)) for r in range(2)]
    for p in procs:
        p.start()
    parts = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    full = {n: np.zeros(g, parts[0][2][n].dtype) for n in _fields(stencil)}
    for _, f, out in parts:
        for n, a in out.items():
            full[n][f[0]:f[0] + a.shape[0], f[1]:f[1] + a.shape[1], f[2]:f[2] + a.shape[2]] = a
    return full


def _single(stencil, g, steps, opts):
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    soln = _run_rank(stencil, g, steps, opts, None)
    _, out = _local_result(soln, stencil, steps)
    soln.end_solution()
    return out


@pytest.mark.parametrize("nr,opts", [((2, 1, 1), ""), ((1, 1, 2), ""), ((1, 2, 1), "-no-overlap_comms"),
                                     ((2, 1, 1), "-no-hip_direct_halo"),      # packed path for the x faces too
                                     ((2, 1, 1), "-min_exterior 12 -hip_variant star25d_z128_y16_r1_u")])
def test_iso3dfd_two_ranks_equal_one_rank(gpu, nr, opts):
    g, steps = (48, 40, 72), 4
    # bit-exactness needs one kernel everywhere: name it (on grids this small prepare_solution() otherwise times the
    # shapes and each rank keeps its own winner) and keep thin y/z exterior slabs on the marching kernel
    if "-hip_variant" not in opts:
        opts += " -hip_variant starlin_v4_z128_y16_r1_m_nt_w2_c4"
    opts = (opts + " -no-hip_thin_slab_point_kernel").strip()
    two = _two_ranks("iso3dfd", g, steps, opts, nr)
    one = _single("iso3dfd", g, steps, opts)
    assert np.array_equal(two["p"], one["p"])
    ref = O.run_iso3dfd(g, steps)[("p", steps)]
    assert O.rel_linf(two["p"], ref) <= 2e-5


def test_ssg_two_ranks_equal_one_rank(gpu):
    """9 in-place fields, 2 stages with an exchange after each, asymmetric halos (3/4), `mu` read
    diagonally (L1 norm 2 -> edge neighbours, here none with 2 ranks, but the boundary extension applies)."""
    g, steps = (40, 24, 36), 3
    opts = "-hip_variant march_v2_z128_y8_w2 -no-hip_thin_slab_point_kernel"        # one kernel everywhere (see above)
    two = _two_ranks("ssg", g, steps, opts, (2, 1, 1))
    one = _single("ssg", g, steps, opts)
    ref = O.run_ssg(g, steps)
    for n in O.SSG_FIELDS:
        assert np.array_equal(two[n], one[n]), n
        r = ref[(n, steps)].astype(np.float64)
        assert np.abs(two[n].astype(np.float64) - r).max() / max(1e-30, np.abs(r).max()) <= 2e-5, n


def test_thin_exterior_slabs_on_the_point_kernel(gpu):
    """Default for y/z decompositions: exterior slabs much thinner than a marching tile are computed by the point
    kernel (different summation order -> compare with the stated tolerance, not bit-exactly)."""
    g, steps = (40, 48, 96), 3
    two = _two_ranks("iso3dfd", g, steps, "-no-hip_planned_launch", (1, 1, 2))       # (the slab schedule: planned launches have no thin slabs)
    ref = O.run_iso3dfd(g, steps)[("p", steps)]
    assert O.rel_linf(two["p"], ref) <= 2e-5


def test_axis3_two_ranks_y_split(gpu):
    g, steps = (24, 44, 40), 3
    two = _two_ranks("3axis", g, steps, "", (1, 2, 1))
    ref = O.run_axis3(g, steps)[("A", steps)]
    assert O.rel_linf(two["A"], ref) <= 1e-12


def test_bench_two_ranks_on_one_gpu(gpu):
    """bench.py's N>1 flow (rank env from torch.distributed.run, x-slab decomposition, barrier-bracketed timing,
    max over ranks, one JSON line from rank 0), with gloo + the host-staged transport so that two ranks can share
    the single GPU of the test box."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, YASK_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(root / "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--size", "128", "--transport", "torch", "--config", "weak", "--ramp-secs", "0.2"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 4 and j["scaling"] == "weak" and j["value"] > 0
    assert j["config"]["decomposition"] == "x-slabs 2x1x1" and "global 256x128x128" in j["config"]["workload"]
    assert j["halo"]["bytes_sent_per_step_rank0"] > 0 and j["step_ms"]["n"] == 4
    # the step's launch schedule was picked by timing every candidate during warm-up (max over ranks), and is reported
    tr = j["config"]["schedule_trials_ms_per_step"]
    assert set(tr) == {"planned", "halves", "serial"} and all(v > 0 for v in tr.values())
    assert j["config"]["schedule"] == min(tr, key=tr.get) and j["config"]["overlap_comms"] == (j["config"]["schedule"] != "serial")
    # the default mode cuts ONE global grid over the ranks (strong scaling, north_star's "1024^3 at 1, 2, 4, 8")
    cmd[cmd.index("--config") + 1] = "c2"
    cmd[cmd.index("--master-port") + 1] = str(_free_port())
    cmd += ["--schedule", "serial"]             # a named schedule is used as given
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["scaling"] == "strong" and "global 128x128x128, 64x128x128 points per GPU" in j["config"]["workload"]
    assert j["config"]["schedule"] == "serial" and j["config"]["overlap_comms"] is False and j["config"]["schedule_trials_ms_per_step"] is None
    assert j["config"]["decomposition"] == "compact rank grid 2x1x1"


def test_bench_eight_ranks_compact_grid_on_one_gpu(gpu):
    """bench.py --gpus 8 as the driver launches it, default configuration (global grid cut over the reference's
    most-compact rank grid, 2x2x2) and BASELINE config 4's shape (--config c4), eight ranks sharing the GPU through gloo +
    the host-staged transport: the JSON line, the decomposition and the halo accounting."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, YASK_DIST_BACKEND="gloo")
    for extra, want_scaling, want_local in ((["--size", "128"], "strong", "64x64x64"), (["--config", "c4", "--size", "64"], "weak", "64x64x32")):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), str(root / "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1",
               "--transport", "torch", "--ramp-secs", "0.1"] + extra
        r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1
        j = json.loads(lines[0])
        assert j["n_gpus"] == 8 and j["scaling"] == want_scaling and j["value"] > 0
        assert j["config"]["decomposition"] == "compact rank grid 2x2x2" and f"{want_local} points per GPU" in j["config"]["workload"]
        assert j["halo"]["bytes_sent_per_step_rank0"] > 0 and j["halo"]["msgs_per_step_rank0"] >= 3      # three face neighbours
