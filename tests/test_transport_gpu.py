"""Halo transports and rank bootstrap on the GPU box (VERDICT r01 missing #2/#4, weak #8):

* the built-in RCCL transport (yask_amd/csrc/ykh_rccl.cpp: dlopen, ncclCommInitRank, grouped ncclSend/ncclRecv,
  ncclAllReduce) executed on ONE GPU as a 1-rank communicator exchanging with itself -- the only way this code can run
  outside a multi-GPU job (RCCL wants one device per rank);
* the native, torch-free bootstrap yk_env_init_from_launcher() (what the C++ yk_factory::new_env() does) with two
  processes sharing the GPU through the host-staged TCP transport: bit-exact against one rank, per-phase timers
  (the reference's halo pack / unpack / wait and exterior / interior times, context.hpp:319-328), and
  exchange_halos() after a change made on ONE rank only;
* run-time knobs added with them: step timers, bandwidth probe, re-allocation after set_alloc_size().
"""
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


def test_rccl_transport_loopback_on_one_gpu(gpu):
    from yask_amd import yk_factory
    fac = yk_factory("iso3dfd")
    env = fac.new_env()
    uid = env.rccl_get_unique_id()
    assert len(uid) == 128 and any(uid)
    env.init_rccl(uid, 0, 1)                     # ncclCommInitRank(nranks = 1)
    assert env.get_num_ranks() == 1 and env.get_rank_index() == 0
    for nbytes in (4096, 1 << 22, 38 * (1 << 20)):   # the last one = an x-face message of the 1024^3 bench
        env.transport_loopback(nbytes)           # group { ncclRecv(self); ncclSend(self) } + ncclAllReduce, bytes verified
    assert env.sum_over_ranks(5) == 5            # rccl_allreduce again, through the env API
    # a solution in that env still runs (no neighbours with one rank)
    soln = fac.new_solution(env)
    soln.set_overall_domain_size_vec([32, 32, 64])
    soln.prepare_solution()
    soln.get_var("p").set_elements_hash(0.0, 1.0, hash_id=0)
    soln.get_var("v").set_elements_hash(150.0, 50.0, hash_id=1)
    soln.run_solution(0, 1)
    ref = O.run_iso3dfd((32, 32, 64), 2)[("p", 2)]
    got = soln.get_var("p").get_elements_in_slice([2, 0, 0, 0], [2, 31, 31, 63])[0]
    assert O.rel_linf(got, ref) <= 2e-5


def test_loopback_without_a_transport_fails_loudly(gpu):
    from yask_amd import yk_factory
    env = yk_factory("iso3dfd").new_env()
    with pytest.raises(RuntimeError, match="no halo transport"):
        env.transport_loopback(4096)


def test_bandwidth_probe_and_step_timers(gpu):
    from yask_amd import yk_factory
    fac = yk_factory("iso3dfd")
    env = fac.new_env()
    copy, mix, rd = (env.probe_bandwidth(k, 1 << 28, 2) for k in (0, 1, 2))
    print(f"probe: copy {copy:.0f} GB/s, 3r1w {mix:.0f} GB/s, read {rd:.0f} GB/s")
    assert 500 < copy < 9000 and 500 < mix < 9000 and 500 < rd < 9000
    soln = fac.new_solution(env)
    soln.set_overall_domain_size_vec([128, 128, 128])
    assert soln.apply_command_line_options("-hip_step_timers -no-auto_tune") == ""
    soln.prepare_solution()
    soln.run_solution(0, 6)
    ms = soln.get_step_times()
    assert len(ms) == 7 and all(0 < x < 100 for x in ms)
    st = soln.get_stats()
    assert st.get_halo_bytes_sent() == 0 and st.get_halo_wait_secs() == 0.0      # single rank: no exchange
    # -no-auto_tune also disables the prepare-time timing pass: two solutions pick the same (static) shape
    s2 = fac.new_solution(env, soln)
    s2.prepare_solution()
    assert s2.get_kernel_variant(0) == soln.get_kernel_variant(0)


def test_step_alloc_change_reallocates(gpu):
    """set_alloc_size(step dim) after a first prepare_solution() must re-allocate (the old check
    compared strides only and kept the 1-slot buffer)."""
    from yask_amd import yk_factory
    fac = yk_factory("iso3dfd")
    soln = fac.new_solution(fac.new_env())
    soln.set_overall_domain_size_vec([16, 16, 32])
    u = soln.new_var("u", ["t", "x", "y", "z"])
    soln.prepare_solution()
    b1 = u.get_num_storage_bytes()
    u.set_alloc_size("t", 4)
    soln.prepare_solution()
    assert u.get_num_storage_bytes() == 4 * b1
    for t in range(4):
        u.set_elements_in_slice_same(float(t + 1), [t, 0, 0, 0], [t, 15, 15, 31], True)
    for t in range(4):        # slot 3 would have been out of bounds of the old allocation
        assert u.get_element([t, 15, 15, 31]) == float(t + 1)


# ------------------------------------------------------------------ several ranks, one GPU, native bootstrap + TCP transport
_FSG = [f"v_{g}_{c}" for g in ("bl", "br", "tl", "tr") for c in "uvw"] + [f"s_{g}_{c}" for g in ("bl", "br", "tl", "tr") for c in ("xx", "yy", "zz", "yz", "xz", "xy")]
FIELDS = {"iso3dfd": ["p"], "ssg": O.SSG_FIELDS, "test_boundary_3d": ["A"], "cube": ["A"], "fsg": _FSG, "fsg_abc": _FSG,
          "awp_abc": ["vel_x", "vel_y", "vel_z", "stress_xx", "stress_yy", "stress_zz", "stress_xy", "stress_xz", "stress_yz"]}
KERNEL = {"iso3dfd": "-hip_variant starlin_v4_z128_y16_r1_m_nt_w2_c4 -no-hip_thin_slab_point_kernel",
          "ssg": "-hip_variant march_v2_z128_y8_w2 -no-hip_thin_slab_point_kernel",      # one kernel everywhere: bit-exact vs 1 rank
          "test_boundary_3d": "-hip_variant naive", "awp_abc": "-hip_variant naive",      # (sub-domain parts: the point kernel)
          "cube": "-hip_variant box_v4_z128_y16_r1_nt_w2 -no-hip_thin_slab_point_kernel",   # (tests/test_box_kernel_gpu.py)
          # (tests/test_clusters_gpu.py: both parts as four clusters of equations on the point kernel; the absorbing parts of
          #  fsg_abc -- box lists per rank -- have no such shape and keep the point kernel)
          "fsg": "-hip_variant c4_vecpt_v4_z256_y4_x1 -no-hip_thin_slab_point_kernel",
          "fsg_abc": "-hip_variant naive"}


def _init(soln, stencil):
    if stencil not in O.DEFAULT_INIT:        # the reference's test stencils: every var hashed, as tests/golden/make_golden.py does
        for i, v in enumerate(soln.get_vars()):
            v.set_elements_hash(1.5, 0.5, hash_id=i)
        return
    init = O.DEFAULT_INIT[stencil]
    for v in soln.get_vars():
        v.set_elements_hash(*init[v.get_name()], hash_id=O.VAR_IDS[stencil][v.get_name()])


def _tcp_worker(rank, world, port, q, mode, stencil, g, nr, steps):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      YASK_HIP_TRANSPORT=os.environ.get("YASK_TEST_TRANSPORT", "tcp"))
    from yask_amd import yk_factory
    fac = yk_factory(stencil)
    env = fac.new_env()
    env.init_from_launcher()                     # no torch.distributed anywhere in this process
    assert env.get_num_ranks() == world and env.get_rank_index() == rank
    env.transport_loopback(1 << 16)
    assert env.sum_over_ranks(rank + 1) == world * (world + 1) // 2
    soln = fac.new_solution(env)
    soln.set_overall_domain_size_vec(list(g))
    if nr is not None:
        soln.set_num_ranks_vec(list(nr))
    assert soln.apply_command_line_options(KERNEL[stencil] + " " + os.environ.get("YASK_TEST_EXTRA_OPTS", "")) == ""
    soln.prepare_solution()
    _init(soln, stencil)
    if mode == "one_sided":
        # only rank 0 changes data, right at the face it shares with rank 1; then EVERY rank calls exchange_halos()
        p = soln.get_var("p")
        f0, l0 = soln.get_first_rank_domain_index_vec(), soln.get_last_rank_domain_index_vec()
        if rank == 0:
            p.set_element(123.5, [1, l0[0], 7, 9])
        soln.exchange_halos()
        q.put((rank, p.get_element([1, f0[0] - 1, 7, 9]) if rank == 1 else None))   # my left halo = rank 0's last plane
        env.global_barrier()
        return
    if mode == "counters":
        # the IPC transport's control plane: registrations travel once per channel, then the host only enqueues device work
        soln.run_solution(0, steps - 1)
        c1 = env.get_transport_counters()
        soln.run_solution(steps, 2 * steps - 1)
        c2 = env.get_transport_counters()
        soln.run_solution(2 * steps, 4 * steps - 1)
        c3 = env.get_transport_counters()
        # buffers re-made (what the reference's prepare_solution() does to its MPI buffers): every rank re-registers, together
        soln.prepare_solution()
        _init(soln, stencil)
        soln.run_solution(0, steps - 1)
        c4 = env.get_transport_counters()
        f, l = soln.get_first_rank_domain_index_vec(), soln.get_last_rank_domain_index_vec()
        out = {n: soln.get_var(n).get_elements_in_slice([steps] + f, [steps] + l)[0] for n in FIELDS[stencil]}
        q.put((rank, f, out, dict(c=[c1, c2, c3, c4], grid=soln.get_num_ranks_vec())))
        env.global_barrier()
        soln.end_solution()
        return
    soln.run_solution(0, steps - 1)
    st = soln.get_stats()
    f, l = soln.get_first_rank_domain_index_vec(), soln.get_last_rank_domain_index_vec()
    out = {n: soln.get_var(n).get_elements_in_slice([steps] + f, [steps] + l)[0] for n in FIELDS[stencil]}
    stats = dict(sent=st.get_halo_bytes_sent(), recv=st.get_halo_bytes_recv(), msgs=st.get_halo_msgs_sent(),
                 pack=st.get_halo_pack_secs(), xfer=st.get_halo_xfer_secs(), unpack=st.get_halo_unpack_secs(),
                 wait=st.get_halo_wait_secs(), ext=st.get_exterior_secs(), inter=st.get_interior_secs(), halo=st.get_halo_secs(),
                 hidden=st.get_comm_hidden_fraction(), grid=soln.get_num_ranks_vec())
    q.put((rank, f, out, stats))
    env.global_barrier()
    soln.end_solution()


def _run_ranks(world, mode, stencil="iso3dfd", g=(48, 40, 72), nr=(2, 1, 1), steps=4):
    import queue
    import time
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tcp_worker, args=(r, world, port, q, mode, stencil, g, nr, steps)) for r in range(world)]
    for p in procs:
        p.start()
    parts, t0 = [], time.time()
    while len(parts) < world:
        try:
            parts.append(q.get(timeout=2))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() - t0 > 300:          # a rank died (its traceback is on stderr): do not wait for the others
                for p in procs:
                    if p.is_alive():
                        p.terminate()
                pytest.fail(f"rank process(es) failed: exit codes {[p.exitcode for p in procs]}")
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return parts


def _one_rank(stencil, g, steps):
    from yask_amd import yk_factory
    fac = yk_factory(stencil)
    one = fac.new_solution(fac.new_env())
    one.set_overall_domain_size_vec(list(g))
    assert one.apply_command_line_options(KERNEL[stencil]) == ""
    one.prepare_solution()
    _init(one, stencil)
    one.run_solution(0, steps - 1)
    return {n: one.get_var(n).get_elements_in_slice([steps, 0, 0, 0], [steps, g[0] - 1, g[1] - 1, g[2] - 1])[0] for n in FIELDS[stencil]}


def _assemble(parts, stencil, g):
    full = {n: np.zeros(g, parts[0][2][n].dtype) for n in FIELDS[stencil]}
    for _, f, out, _ in parts:
        for n, a in out.items():
            full[n][f[0]:f[0] + a.shape[0], f[1]:f[1] + a.shape[1], f[2]:f[2] + a.shape[2]] = a
    return full


@pytest.mark.parametrize("transport", ["tcp", "ipc"])
@pytest.mark.parametrize("mode", ["x", "z"])
def test_native_bootstrap_two_ranks_equal_one_rank(gpu, mode, transport, monkeypatch):
    """Two processes, one GPU, no torch: the host-staged TCP transport and the device-to-device IPC transport (HIP IPC
    handles, copies into the neighbour's buffer, stream-ordered flag words: yask_amd/csrc/ykh_ipc.cpp)."""
    monkeypatch.setenv("YASK_TEST_TRANSPORT", transport)
    g, steps = (48, 40, 72), 4
    parts = _run_ranks(2, mode, nr=(2, 1, 1) if mode == "x" else (1, 1, 2))
    full = _assemble(parts, "iso3dfd", g)["p"]
    assert np.array_equal(full, _one_rank("iso3dfd", g, steps)["p"])
    assert O.rel_linf(full, O.run_iso3dfd(g, steps)[("p", steps)]) <= 2e-5
    # per-phase accounting: one face, 8 planes (x) or 8 columns (z) of one var slot per step, both directions
    for _, _, _, s in parts:
        face = (40 * 72 if mode == "x" else 48 * 40) * 8 * 4
        # run_solution() first exchanges everything that may be dirty (both step slots of p), then one slot per step
        if mode == "x":      # in-place x-face transfer: whole planes WITH their y/z pads
            assert s["sent"] >= (steps + 2) * face and s["recv"] == s["sent"]
        else:
            assert s["sent"] == (steps + 2) * face and s["recv"] == s["sent"]
        # (z cut, default schedule: the pipelined half-exchanges send a face in two messages per step when the box is long enough in x)
        assert steps + 1 <= s["msgs"] <= (steps + 2 if mode == "x" else 2 * steps + 2)
        # (planned launches: "exterior" ends when the comm stream has seen the shell blocks' signal, which on a grid this small
        #  may be after the launch itself has ended -- the interior span is then zero)
        assert s["xfer"] > 0 and s["inter"] >= 0 and s["ext"] > 0 and s["wait"] >= 0
        assert s["halo"] >= s["pack"] + s["xfer"] + s["unpack"] - 1e-9
        assert s["hidden"] is not None and 0.0 <= s["hidden"] <= 1.0


@pytest.mark.parametrize("stencil,g,world,nr,opts", [("iso3dfd", (48, 40, 72), 2, (2, 1, 1), ""), ("iso3dfd", (48, 40, 72), 8, (2, 2, 2), "-hip_halves"),
                                                    ("ssg", (32, 28, 40), 8, (2, 2, 2), "")])
def test_ipc_transport_keeps_the_host_out_of_the_exchange(gpu, stencil, g, world, nr, opts, monkeypatch):
    """the address of a receive buffer crosses the TCP mesh ONCE per channel (first use); after
    that run_solution() costs one 8-byte all-reduce per CALL and no control message per step or exchange -- counted by the
    transport itself (yk_env_get_transport_counters).  Re-made buffers (prepare_solution() again) are re-registered by all
    ranks together, and the result after that still equals the one-rank run bit for bit."""
    monkeypatch.setenv("YASK_TEST_TRANSPORT", "ipc")
    monkeypatch.setenv("YASK_TEST_EXTRA_OPTS", opts)
    steps = 4
    parts = _run_ranks(world, "counters", stencil=stencil, g=g, nr=nr, steps=steps)
    for _, _, _, s in parts:
        c1, c2, c3, c4 = s["c"]
        assert s["grid"] == list(nr)
        assert c1["ctl_msgs"] > 0 and c1["ctl_bytes"] == 96 * c1["ctl_msgs"]
        assert c2["ctl_msgs"] == c1["ctl_msgs"] and c3["ctl_msgs"] == c1["ctl_msgs"], (c1, c2, c3)       # steady state: 0 bytes per step
        assert c2["begins"] == c1["begins"] + 1 and c3["begins"] == c2["begins"] + 1                      # one all-reduce per call
        assert c3["resets"] == c1["resets"]
        assert c3["dev_ops"] - c2["dev_ops"] > c2["dev_ops"] - c1["dev_ops"] > 0                          # device work scales with the steps, only
        assert c4["resets"] == c3["resets"] + 1 and c4["ctl_msgs"] == 2 * c1["ctl_msgs"]                  # registered again, once
        assert c1["mailbox_kind"] in (0, 1, 2, 3)
    full = _assemble(parts, stencil, g)
    one = _one_rank(stencil, g, steps)
    for n in FIELDS[stencil]:
        assert np.array_equal(full[n], one[n]), n


@pytest.mark.parametrize("transport", ["tcp", "ipc"])
def test_exchange_halos_after_a_change_on_one_rank_only(gpu, transport, monkeypatch):
    """rank 0 alone marks a var dirty; all ranks call exchange_halos().  Every rank must post the same
    messages (the reference's set_all_neighbor_vars_dirty, context.cpp:234) -- this used to hang / mismatch."""
    monkeypatch.setenv("YASK_TEST_TRANSPORT", transport)
    parts = dict((r, v) for r, v in _run_ranks(2, "one_sided"))
    assert parts[1] == 123.5


# the schedules the library compiles (round 5: four; the one-launch / device-signal, in-line pack, first-planner and side-by-side slab
# variants of rounds 2-3 are deleted): halves (default), planned, slabs + interior, and the exchange after the launch
SCHEDULES = {"default_halves_ipc": "", "planned": "-no-hip_halves", "planned_ipc": "-no-hip_halves", "slabs": "-no-hip_planned_launch",
             "serial_ipc": "-no-overlap_comms"}


@pytest.mark.parametrize("stencil,g,steps,sched", [("iso3dfd", (48, 40, 72), 3, k) for k in SCHEDULES] +
                         [("ssg", (32, 28, 40), 2, k) for k in ("default_halves_ipc", "planned", "slabs", "serial_ipc")])
def test_eight_ranks_on_the_compact_2x2x2_grid(gpu, stencil, g, steps, sched, monkeypatch):
    """BASELINE.json configs[3]/[4] run on the reference's default rank grid for 8 ranks, 2x2x2
    (get_compact_factors, src/common/tuple.cpp:355-430): 3 face neighbours per rank for iso3dfd; ssg's `mu` is read
    diagonally (L1 norm 2), so its halos also travel to the 3 edge neighbours.  Eight processes share the GPU (TCP
    or IPC transport); the assembled result equals the 1-rank run bit for bit under every schedule the library has: the pipelined
    half-exchanges (default; boxes too short in x fall back to planned launches), ONE plan of equal blocks per stage with the shell
    blocks first (-no-hip_halves), exterior slabs + interior (-no-hip_planned_launch), and the exchange after the launch."""
    monkeypatch.setenv("YASK_TEST_EXTRA_OPTS", SCHEDULES[sched])
    monkeypatch.setenv("YASK_TEST_TRANSPORT", "ipc" if sched.endswith("_ipc") else "tcp")
    parts = _run_ranks(8, "run", stencil=stencil, g=g, nr=None, steps=steps)
    assert all(s["grid"] == [2, 2, 2] for _, _, _, s in parts)
    full = _assemble(parts, stencil, g)
    one = _one_rank(stencil, g, steps)
    for n in FIELDS[stencil]:
        assert np.array_equal(full[n], one[n]), n
    ref = O.run_iso3dfd(g, steps) if stencil == "iso3dfd" else O.run_ssg(g, steps)
    for n in FIELDS[stencil]:
        r = ref[(n, steps)].astype(np.float64)
        assert np.abs(full[n].astype(np.float64) - r).max() / max(1.0 if stencil == "iso3dfd" else 1e-30, np.abs(r).max()) <= 2e-5, n


@pytest.mark.parametrize("stencil,g,world,nr,steps,transport", [("iso3dfd", (64, 40, 72), 8, (2, 2, 2), 4, "ipc"), ("ssg", (64, 28, 40), 8, (2, 2, 2), 3, "ipc"),
                                                               ("iso3dfd", (40, 48, 40), 4, (1, 2, 2), 3, "tcp"), ("iso3dfd", (72, 40, 72), 2, (1, 1, 2), 5, "ipc")])
def test_pipelined_half_exchanges_equal_one_rank(gpu, stencil, g, world, nr, steps, transport, monkeypatch):
    """-hip_halves: a stage of a decomposed rank is two launches in regular order -- the outer x-half [0, nx/4) u [3nx/4, nx), then
    the inner half -- and each is followed by the exchange of ITS part of the faces, finished only after the other half has been
    launched (the reference progresses its messages while it computes: adv_halo_exchange, src/kernel/lib/halo.cpp:494-574).  Same
    kernels, same per-point arithmetic, halos in place before the launch that reads them: the assembled result equals the one-rank
    run bit for bit -- 2x2x2 (x faces in place with the outer half, y / z faces cut in x), grids without x neighbours, two
    stages with in-place fields (ssg), odd and even step counts, the device-to-device and the host-staged transport."""
    monkeypatch.setenv("YASK_TEST_TRANSPORT", transport)
    monkeypatch.setenv("YASK_TEST_EXTRA_OPTS", "-hip_halves")
    parts = _run_ranks(world, "run", stencil=stencil, g=g, nr=nr, steps=steps)
    full = _assemble(parts, stencil, g)
    one = _one_rank(stencil, g, steps)
    for n in FIELDS[stencil]:
        assert np.array_equal(full[n], one[n]), n
    # the schedule really ran (a box too short in x falls back to planned launches): per step and stage every y / z face neighbour
    # gets TWO messages, an x neighbour one (every rank of these grids is a corner: one neighbour per decomposed dim)
    nx_nb = 1 if nr[0] > 1 else 0
    nyz_nb = sum(1 for d in (1, 2) if nr[d] > 1)
    for _, _, _, st in parts:
        assert st["grid"] == list(nr)
        if stencil == "iso3dfd":
            # (run_solution() first exchanges both step slots of p: in-place x faces are one message per slot)
            # (... with the host-staged transport; the IPC transport packs x faces like the others: one message per neighbour)
            first = (nx_nb if transport == "ipc" else 2 * nx_nb) + nyz_nb
            assert st["msgs"] == first + steps * (nx_nb + 2 * nyz_nb), st["msgs"]
        else:
            assert st["msgs"] >= 2 * steps * (nx_nb + 2 * nyz_nb), st["msgs"]
        assert st["wait"] >= 0 and st["ext"] > 0 and st["xfer"] > 0 and st["hidden"] is not None


@pytest.mark.parametrize("transport", ["tcp", "ipc"])
@pytest.mark.parametrize("nr,g", [((4, 1, 1), (64, 24, 40)), ((2, 2, 1), (40, 48, 40)), ((1, 2, 2), (24, 40, 72))])
def test_four_ranks_slab_and_pencil_grids(gpu, nr, g, transport, monkeypatch):
    """Four ranks: x-slabs (the two middle ranks have an in-place x-face transfer on both sides), and the 2x2x1 / 1x2x2
    grids (one direct and one packed face per rank; 1x2x2: packed faces only, thin y/z exteriors)."""
    monkeypatch.setenv("YASK_TEST_TRANSPORT", transport)
    steps = 3
    parts = _run_ranks(4, "run", stencil="iso3dfd", g=g, nr=nr, steps=steps)
    assert all(s["grid"] == list(nr) for _, _, _, s in parts)
    full = _assemble(parts, "iso3dfd", g)["p"]
    assert np.array_equal(full, _one_rank("iso3dfd", g, steps)["p"])
    assert O.rel_linf(full, O.run_iso3dfd(g, steps)[("p", steps)]) <= 2e-5


@pytest.mark.parametrize("stencil,g,world,nr,steps", [("iso3dfd", (48, 40, 72), 2, (2, 1, 1), 5), ("iso3dfd", (48, 40, 72), 2, (1, 1, 2), 4),
                                                     ("iso3dfd", (48, 40, 72), 8, None, 4), ("ssg", (40, 36, 40), 8, None, 4),
                                                     ("ssg", (40, 36, 40), 2, (1, 2, 1), 3)])
def test_wave_front_tiling_across_ranks_halves_the_exchanges(gpu, stencil, g, world, nr, steps, monkeypatch):
    """-Mbt 2 with neighbours (VERDICT r02 missing #2; the reference's wave-fronts across ranks, setup.cpp:717-805,
    context.cpp:286-346): halos grow by (2 x stages - 1) x the stencil halo, every rank evaluates the phases of a two-step
    group on boxes that shrink towards its own -- redundantly with its neighbours -- and halos travel ONCE per group, to all
    26 neighbours (extended boxes have edges and corners).  Same kernels, same per-point arithmetic: the assembled result
    equals the one-rank run bit for bit (odd step counts leave a one-step group at the end), with half the exchanges."""
    monkeypatch.setenv("YASK_TEST_TRANSPORT", "ipc")
    monkeypatch.setenv("YASK_TEST_EXTRA_OPTS", "-Mbt 2")
    wf = _run_ranks(world, "run", stencil=stencil, g=g, nr=nr, steps=steps)
    monkeypatch.setenv("YASK_TEST_EXTRA_OPTS", "")
    plain = _run_ranks(world, "run", stencil=stencil, g=g, nr=nr, steps=steps)
    full = _assemble(wf, stencil, g)
    one = _one_rank(stencil, g, steps)
    for n in FIELDS[stencil]:
        assert np.array_equal(full[n], one[n]), n
    stages = 2 if stencil == "ssg" else 1
    for (_, _, _, a), (_, _, _, b) in zip(sorted(wf, key=lambda x: x[0]), sorted(plain, key=lambda x: x[0])):
        # plain sweeps: one exchange per stage per step (+ the initial one), with the face neighbours (ssg: `mu` also goes to the
        # edge neighbours, once); wave-fronts: one exchange per group of two steps (+ the initial one), with every neighbour of
        # the 26-neighbourhood that exists -- 7 instead of 3 for a rank of the 2x2x2 grid, 1 instead of 1 with two ranks
        n26 = 7 if world == 8 else 1
        groups = (steps + 1) // 2
        assert a["msgs"] == (groups + 1) * n26, (a["msgs"], groups, n26)
        assert b["msgs"] >= steps * stages * (3 if world == 8 else 1)
        assert a["msgs"] / n26 < b["msgs"] / (3 if world == 8 else 1)          # fewer rounds of communication
        assert a["sent"] > 0


@pytest.mark.parametrize("stencil,g,world,nr,steps", [("test_boundary_3d", (40, 36, 48), 2, (2, 1, 1), 4), ("test_boundary_3d", (40, 36, 48), 2, (1, 1, 2), 3),
                                                     ("test_boundary_3d", (40, 36, 48), 8, None, 4), ("awp_abc", (48, 40, 56), 2, (1, 2, 1), 2)])
def test_wave_front_tiling_across_ranks_with_sub_domain_parts(gpu, stencil, g, world, nr, steps, monkeypatch):
    """with -Mbt 2 and neighbours the early phases of a group run on boxes grown into the neighbours'
    domains; parts under a domain condition (IF_DOMAIN: test_boundary_3d's two half-space equations, awp_abc's sponge layers and
    free surface) are launched over the bounding box of their condition, which therefore has to be found over the GROWN box (the
    reference: find_bounding_boxes over ext_bb, setup.cpp:1000-1075) -- clipped to the rank box the parts were skipped out there
    and the next phase read stale values near the rank boundary.  The assembled result equals the one-rank run bit for bit."""
    monkeypatch.setenv("YASK_TEST_TRANSPORT", "ipc")
    monkeypatch.setenv("YASK_TEST_EXTRA_OPTS", "-Mbt 2")
    wf = _run_ranks(world, "run", stencil=stencil, g=g, nr=nr, steps=steps)
    full = _assemble(wf, stencil, g)
    one = _one_rank(stencil, g, steps)
    for n in FIELDS[stencil]:
        assert np.isfinite(one[n]).all()
        assert np.array_equal(full[n], one[n]), (n, float(np.abs(full[n] - one[n]).max()), np.argwhere(full[n] != one[n])[:4].tolist())
    groups = (steps + 1) // 2
    for _, _, _, a in wf:
        assert a["msgs"] == (groups + 1) * (7 if world == 8 else 1), a["msgs"]      # the wave-front schedule really ran
