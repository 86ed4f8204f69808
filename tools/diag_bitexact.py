#!/usr/bin/env python
"""GPU diagnostic: where (if anywhere) does a decomposed / re-chunked iso3dfd run differ from the plain one-rank run?"""
import os
import sys
import socket
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import oracle as O  # noqa: E402

G = (1024, 1024, 1024)
STEPS = 2
PLANES = list(range(0, 16)) + list(range(248, 264)) + list(range(496, 528)) + list(range(760, 776)) + list(range(1008, 1024))


def init(soln):
    for v in soln.get_vars():
        v.set_elements_hash(*O.DEFAULT_INIT["iso3dfd"][v.get_name()], hash_id=O.VAR_IDS["iso3dfd"][v.get_name()])


def run_one(opts):
    from yask_amd import yk_factory
    fac = yk_factory("iso3dfd")
    s = fac.new_solution(fac.new_env())
    s.set_overall_domain_size_vec(list(G))
    assert s.apply_command_line_options(opts) == ""
    s.prepare_solution()
    init(s)
    s.run_solution(0, STEPS - 1)
    p = s.get_var("p")
    out = {x: p.get_elements_in_slice([STEPS, x, 0, 0], [STEPS, x, G[1] - 1, G[2] - 1])[0][0].copy() for x in PLANES}
    kern = s.get_kernel_variant(0)
    s.end_solution()
    return out, kern


def worker(rank, world, port, q, nr, opts, steps, planes=None):
    global STEPS, PLANES
    STEPS = steps
    if planes is not None:
        PLANES = planes
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), YASK_HIP_TRANSPORT="tcp")
    from yask_amd import yk_factory
    fac = yk_factory("iso3dfd")
    env = fac.new_env()
    env.init_from_launcher()
    s = fac.new_solution(env)
    s.set_overall_domain_size_vec(list(G))
    s.set_num_ranks_vec(list(nr))
    assert s.apply_command_line_options(opts) == ""
    s.prepare_solution()
    init(s)
    s.run_solution(0, STEPS - 1)
    f, l = s.get_first_rank_domain_index_vec(), s.get_last_rank_domain_index_vec()
    p = s.get_var("p")
    out = {x: p.get_elements_in_slice([STEPS, x, f[1], f[2]], [STEPS, x, l[1], l[2]])[0][0].copy() for x in PLANES if f[0] <= x <= l[0]}
    q.put((rank, f, l, out))
    env.global_barrier()
    s.end_solution()


def diff(name, a, b, yo=0, zo=0):
    tot, bad, worst = 0, 0, 0.0
    where = []
    for x in sorted(a):
        if x not in b:
            continue
        pa = a[x][yo:yo + b[x].shape[0], zo:zo + b[x].shape[1]]
        d = pa != b[x]
        tot += d.size
        n = int(d.sum())
        if n:
            bad += n
            worst = max(worst, float(np.abs(pa.astype(np.float64) - b[x]).max()))
            ys, zs = np.nonzero(d)
            where.append((x, n, int(ys.min()), int(ys.max()), int(zs.min()), int(zs.max())))
            if len(where) <= 2:
                print("      x=%d first differing (y, z):" % x, list(zip(ys[:24].tolist(), zs[:24].tolist())), "y%32:", sorted(set((ys % 32).tolist()))[:40], "z%128:", sorted(set((zs % 128).tolist()))[:40])
    print(f"{name}: {bad} of {tot} points differ, max |diff| {worst:.3e}")
    for w in where[:6]:
        print("    plane x=%d: %d differ, y in [%d, %d], z in [%d, %d]" % w)


def main():
    import multiprocessing as mp
    global STEPS
    if len(sys.argv) > 1:
        STEPS = int(sys.argv[1])
    a, ka = run_one("")
    print("one rank default kernel", ka)
    a2, _ = run_one("")
    diff("one rank, run twice", a, a2)
    b, _ = run_one("-hip_xchunk 128")
    diff("one rank, x-chunks of 128", a, b)
    c, _ = run_one("-hip_xchunk 512 -no-hip_round_launches")
    diff("one rank, x-chunks of 512, no round launches", a, c)
    cases = (((1, 1, 2), "-no-overlap_comms"), ((1, 1, 2), "-no-hip_planned_launch -no-hip_thin_slab_point_kernel"), ((1, 1, 2), ""),
             ((2, 1, 1), "-no-overlap_comms"), ((1, 2, 1), "-no-overlap_comms"))
    if len(sys.argv) > 2:
        L = "-no-hip_planned_launch -no-hip_thin_slab_point_kernel"
        cases = (((1, 1, 2), L), ((1, 1, 2), L + " -hip_overlap_splits 1"), ((1, 1, 2), L + " -no-hip_round_launches"),
                 ((1, 1, 2), L + " -hip_overlap_splits 1 -no-hip_round_launches"), ((1, 1, 2), L + " -hip_xchunk 1024 -hip_overlap_splits 1 -no-hip_round_launches"))
    for nr, opts in cases:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        procs = [ctx.Process(target=worker, args=(r, 2, port, q, nr, opts, STEPS)) for r in range(2)]
        for p in procs:
            p.start()
        parts = [q.get(timeout=300) for _ in procs]
        for p in procs:
            p.join(timeout=60)
        for rank, f, l, out in sorted(parts):
            diff(f"two ranks {nr} '{opts}': rank {rank} box {f}..{l}", a, out, f[1], f[2])


if __name__ == "__main__":
    main()
