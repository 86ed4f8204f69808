#!/usr/bin/env python
"""GPU box (one GPU): how much of the halo exchange does each step schedule hide?

One process plays a rank of a decomposed job through the mirror transport (yk_env_init_mirror: what it sends to a neighbour comes
back as what it expects from that neighbour -- a device-to-device copy on the communication stream, the stand-in for an equally
fast peer; halo DATA are those of a reflecting boundary, so this measures time, not values).  The full schedule runs: planned
launch (or slabs + interior, or the whole box), device-side signal, pack kernels, copy, unpack kernels, the compute stream's wait.
Reported per step (HIP events of yk_stats): exterior / interior / pack / transport / unpack / exposed wait, ms per step, and
the undivided single-rank step for reference.
    python tools/overlap_probe.py [--stencil iso3dfd]"""
import argparse
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

CASES = {  # name -> (global size, rank grid, rank played)
    "iso3dfd": [("c2 / 8 GPUs 2x2x2: 512^3 block, 3 face neighbours", (1024, 1024, 1024), (2, 2, 2), 0),
                ("c4 / 8 GPUs 2x2x2: 1024x1024x512 block", (2048, 2048, 1024), (2, 2, 2), 0),
                ("c2 / 4 GPUs 2x2x1: 512x512x1024 block", (1024, 1024, 1024), (2, 2, 1), 0)],
    "ssg": [("ssg 1024^3 / 8 GPUs 2x2x2: 512^3 block", (1024, 1024, 1024), (2, 2, 2), 0),
            ("ssg 512^3 / 8 GPUs 2x2x2: 256^3 block", (512, 512, 512), (2, 2, 2), 0)],
}
# Which rank grid for 8 GPUs?  The reference's default is the most compact one (2x2x2, get_compact_factors, src/common/tuple.cpp:355-430);
# `-nr*` lets the user choose.  Grids that do not cut z have no z face (32-byte runs: the expensive pack / unpack, the column of
# shell tiles) but more halo bytes; x faces travel in place.  The rank played is the one with the most neighbours.
GRID_CASES = [("c2 1024^3 / 2x2x2: 512^3 block", (1024, 1024, 1024), (2, 2, 2), 0),
              ("c2 1024^3 / 2x4x1: 512x256x1024 block, y both sides", (1024, 1024, 1024), (2, 4, 1), 2),
              ("c2 1024^3 / 4x2x1: 256x512x1024 block, x both sides", (1024, 1024, 1024), (4, 2, 1), 1),
              ("c2 1024^3 / 1x4x2: 1024x256x512 block", (1024, 1024, 1024), (1, 4, 2), 1),
              ("c2 1024^3 / 8x1x1: 128x1024x1024 block, x both sides", (1024, 1024, 1024), (8, 1, 1), 1),
              ("c4 2048x2048x1024 / 2x2x2: 1024x1024x512 block", (2048, 2048, 1024), (2, 2, 2), 0),
              ("c4 2048x2048x1024 / 2x4x1: 1024x512x1024 block", (2048, 2048, 1024), (2, 4, 1), 2),
              ("c4 2048x2048x1024 / 4x2x1: 512x1024x1024 block", (2048, 2048, 1024), (4, 2, 1), 1),
              ("c4 2048x2048x1024 / 8x1x1: 256x2048x1024 block", (2048, 2048, 1024), (8, 1, 1), 1)]
SCHEDULES = [("halves: two launches in regular order, half-exchanges pipelined (-hip_halves)", "-overlap_comms -hip_planned_launch -hip_halves"),
             ("planned (rounds, shell first)", "-overlap_comms -hip_planned_launch -no-hip_halves"),
             ("slabs + interior (round 2)", "-overlap_comms -no-hip_planned_launch -no-hip_halves"),
             ("whole box, then exchange", "-no-overlap_comms")]
# the reference's wave-front tiling across ranks (-Mbt n): halos (2 x stages - 1) x wider, every rank evaluates the steps of
# a group on shrinking boxes, ONE exchange per n steps with all neighbours of the 26-neighbourhood.  Changes the allocation (pads): these
# get a solution of their own (--mbt).  SURVEY section 8 row f2 / BASELINE config 5: under which link does it beat plain sweeps?
MBT_SCHEDULES = [("wave-front tiling across ranks, -Mbt 2 (one exchange per 2 steps)", "-overlap_comms -Mbt 2"),
                 ("wave-front tiling across ranks, -Mbt 3", "-overlap_comms -Mbt 3")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stencil", default="iso3dfd")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--cases", type=int, default=99, help="only the first N cases")
    ap.add_argument("--schedules", default="", help="only the schedules whose label contains this text (comma-separated alternatives)")
    ap.add_argument("--tag", default="")
    ap.add_argument("--passes", type=int, default=2, help="interleaved passes over the schedules (they share one solution)")
    ap.add_argument("--fresh-solutions", action="store_true", help="one env + solution per schedule (one pass)")
    ap.add_argument("--mbt", action="store_true", help="also the -Mbt 2 / 3 wave-front schedules, each on a solution of its own")
    ap.add_argument("--grid-study", action="store_true", help="iso3dfd: the 8-GPU rank grids of GRID_CASES instead of the default cases")
    args = ap.parse_args()
    from yask_amd import yk_factory
    from yask_amd.kernel import yk_env
    yk_env.disable_debug_output()
    fac = yk_factory(args.stencil)
    out = []
    for name, g, nr, rank in (GRID_CASES if args.grid_study else CASES[args.stencil])[:args.cases]:
        world = nr[0] * nr[1] * nr[2]
        local = [g[d] // nr[d] for d in range(3)]
        # reference: the same block as a one-rank job
        s = fac.new_solution(fac.new_env())
        s.set_overall_domain_size_vec(local)
        assert s.apply_command_line_options("-no-auto_tune") == ""
        s.prepare_solution()
        for k, v in enumerate(s.get_vars()):
            v.set_elements_hash(1.0, 0.1, hash_id=k)
        s.run_solution(0, 9)
        t0 = time.perf_counter()
        s.run_solution(10, 10 + args.steps - 1)
        one_ms = (time.perf_counter() - t0) / args.steps * 1e3
        s.end_solution()
        # Every schedule is a set of run-time options, so by default all schedules of a case run on ONE env + solution, i.e. on one
        # set of var allocations, in --passes interleaved passes (where the arrays lie is worth 3-4 % of a step, DESIGN.md section 2:
        # more than some schedules differ by); --fresh-solutions = one solution per schedule, as before the end of round 3.
        RESET = "-overlap_comms -hip_planned_launch -no-hip_halves "

        def make(opts):
            env = fac.new_env()
            env.init_mirror(rank, world)
            so = fac.new_solution(env)
            so.set_overall_domain_size_vec(list(g))
            so.set_num_ranks_vec(list(nr))
            assert so.apply_command_line_options("-no-auto_tune " + opts) == ""
            so.prepare_solution()
            for k, v in enumerate(so.get_vars()):
                v.set_elements_hash(1.0, 0.1, hash_id=k)
            return so

        chosen = [x for x in SCHEDULES if any(k in x[0] for k in args.schedules.split(","))]
        shared, t_next = None, 0
        for pass_no, (label, opts) in [(p_, x) for p_ in range(1 if args.fresh_solutions else max(1, args.passes)) for x in chosen]:
            if args.fresh_solutions:
                s, t_next = make(opts), 0
            else:
                if shared is None:
                    shared = make(RESET)
                s = shared
                assert s.apply_command_line_options(RESET + opts) == ""
                for k, v in enumerate(s.get_vars()):         # (fresh values: the test coefficients make the scheme grow without bound)
                    v.set_elements_hash(1.0, 0.1, hash_id=k)
            s.run_solution(t_next, t_next + 9)
            s.get_stats()
            t0 = time.perf_counter()
            s.run_solution(t_next + 10, t_next + 10 + args.steps - 1)
            ms = (time.perf_counter() - t0) / args.steps * 1e3
            t_next += 10 + args.steps
            st = s.get_stats()
            n = args.steps
            comm = st.get_halo_pack_secs() + st.get_halo_xfer_secs() + st.get_halo_unpack_secs()
            rec = {"tag": args.tag, "case": name, "schedule": label, "ms_per_step": round(ms, 4), "one_rank_block_ms_per_step": round(one_ms, 4),
                   "vs_one_rank_block": round(ms / one_ms, 3),
                   "exterior_ms": round(st.get_exterior_secs() / n * 1e3, 4), "interior_ms": round(st.get_interior_secs() / n * 1e3, 4),
                   "pack_ms": round(st.get_halo_pack_secs() / n * 1e3, 4), "copy_ms": round(st.get_halo_xfer_secs() / n * 1e3, 4),
                   "unpack_ms": round(st.get_halo_unpack_secs() / n * 1e3, 4), "exposed_wait_ms": round(st.get_halo_wait_secs() / n * 1e3, 4),
                   "comm_hidden_fraction": round(max(0.0, 1.0 - st.get_halo_wait_secs() / comm), 3) if comm > 0 else None,
                   "halo_MB_per_step": round(st.get_halo_bytes_sent() / n / 1e6, 2), "rank_grid": list(nr), "rank": rank,
                   "job_gpoints_per_s_at_8_ranks": round(g[0] * g[1] * g[2] / ms * 1e-6, 1),
                   "solution": "own" if args.fresh_solutions else "shared by the case's schedules", "pass": pass_no}
            out.append(rec)
            print(json.dumps(rec), flush=True)
            if args.fresh_solutions:
                s.end_solution()
        if shared is not None:
            shared.end_solution()
        for label, opts in (MBT_SCHEDULES if args.mbt else []):
            try:
                s = make(opts)
                s.run_solution(0, 11)
                s.get_stats()
                n = args.steps // 6 * 6           # whole groups of 2 and of 3 steps
                t0 = time.perf_counter()
                s.run_solution(12, 12 + n - 1)
                ms = (time.perf_counter() - t0) / n * 1e3
                st = s.get_stats()
                rec = {"tag": args.tag, "case": name, "schedule": label, "ms_per_step": round(ms, 4), "one_rank_block_ms_per_step": round(one_ms, 4),
                       "vs_one_rank_block": round(ms / one_ms, 3), "exposed_wait_ms": round(st.get_halo_wait_secs() / n * 1e3, 4),
                       "halo_MB_per_step": round(st.get_halo_bytes_sent() / n / 1e6, 2), "msgs_per_step": round(st.get_halo_msgs_sent() / n, 2),
                       "rank_grid": list(nr), "rank": rank, "solution": "own"}
                out.append(rec)
                print(json.dumps(rec), flush=True)
                s.end_solution()
            except RuntimeError as ex:
                print(json.dumps({"case": name, "schedule": label, "error": str(ex)[:300]}), flush=True)
    od = Path(__file__).resolve().parents[1] / "gpurun_out"
    od.mkdir(exist_ok=True)
    json.dump(out, open(od / f"overlap_probe_{args.stencil}{args.tag}.json", "w"), indent=1)


if __name__ == "__main__":
    main()
