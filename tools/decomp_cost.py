#!/usr/bin/env python
"""GPU box (one GPU): what does the exterior-first schedule of a decomposed run cost on the compute side?

For the local block a rank gets in bench.py's scaling configurations, replays the launches Solution::run() issues for
one step, with no communication, and compares with the undivided box (yk_solution_time_decomposed_step):
  * planned launches (round 3, the default): ONE launch of the marching kernel over the rank box, shell blocks first; `shell_ms`
    = until the device-side signal that releases the halo exchange, `rest_ms` = from there to the end of the launch;
  * round 2's separate launches (-no-hip_planned_launch): exterior slabs (thin ones on the point kernel, the z exterior one
    marching tile wide), then the interior.
    python tools/decomp_cost.py [--stencil iso3dfd] [--quick]"""
import argparse
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

CASES = [  # name, local size, neighbours on the (lo, hi) side of x, y, z
    ("c2 / 8 GPUs, 2x2x2, local 512^3 (corner rank: one neighbour per dim)", (512, 512, 512), (0, 0, 0), (1, 1, 1)),
    ("c4 / 8 GPUs, 2x2x2, local 1024x1024x512", (1024, 1024, 512), (0, 0, 0), (1, 1, 1)),
    ("c2 / 4 GPUs, 2x2x1, local 512x512x1024", (512, 512, 1024), (0, 0, 0), (1, 1, 0)),
    ("c2 / 2 GPUs, 2x1x1, local 512x1024x1024", (512, 1024, 1024), (0, 0, 0), (1, 0, 0)),
    ("c2 / 8 GPUs as x-slabs, middle rank, local 128x1024x1024", (128, 1024, 1024), (1, 0, 0), (1, 0, 0)),
]
SSG_CASES = [
    ("ssg 512^3 global / 8 GPUs, 2x2x2, local 256^3", (256, 256, 256), (0, 0, 0), (1, 1, 1)),
    ("ssg 1024^3 global / 8 GPUs, 2x2x2, local 512^3", (512, 512, 512), (0, 0, 0), (1, 1, 1)),
]
CONFIGS = [  # label, options
    # the schedules the library compiles since round 5 (the first planner, shell percentages, in-line pack, the one-launch / device-signal
    # form, split interiors and side-by-side slabs of rounds 2-3 were measured here -- profiles/r3_decomp, r3_halves -- and are deleted)
    ("halves: two launches in regular order (default)", "-hip_planned_launch -hip_halves"),
    ("planned: one plan of equal blocks, shell first (-no-hip_halves)", "-no-hip_halves -hip_planned_launch"),
    ("slabs + interior (-no-hip_planned_launch)", "-no-hip_halves -no-hip_planned_launch"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stencil", default="iso3dfd")
    ap.add_argument("--quick", action="store_true", help="the two BASELINE blocks, three configurations")
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--configs", default="", help="only the configurations whose label contains this text (comma-separated alternatives)")
    ap.add_argument("--cases", type=int, default=99, help="only the first N cases")
    ap.add_argument("--passes", type=int, default=2, help="interleaved passes over the configurations that share a solution")
    ap.add_argument("--fresh-solutions", action="store_true", help="one solution (one set of allocations) per configuration, as before round 3's end")
    args = ap.parse_args()
    from yask_amd import yk_factory
    from yask_amd.kernel import yk_env
    yk_env.disable_debug_output()
    fac = yk_factory(args.stencil)
    cases = SSG_CASES if args.stencil == "ssg" else CASES
    configs = CONFIGS
    if args.quick:
        cases, configs = cases[:2], [CONFIGS[0], CONFIGS[1], CONFIGS[3], CONFIGS[4], CONFIGS[8], CONFIGS[9]]
    if args.configs:
        configs = [c for c in configs if any(k in c[0] for k in args.configs.split(","))]
    cases = cases[:args.cases]
    out = []
    ramped = [False]

    def make(size, opts):
        s = fac.new_solution(fac.new_env())
        s.set_overall_domain_size_vec(list(size))
        assert s.apply_command_line_options("-no-auto_tune " + opts) == ""
        s.prepare_solution()
        if not ramped[0]:          # a box that idled is at low clocks: ~1.5 s of plain steps before the first measurement
            import time
            t0, t = time.perf_counter(), 0
            while time.perf_counter() - t0 < 1.5:
                s.run_solution(t, t + 19)
                t += 20
            ramped[0] = True
        for k, v in enumerate(s.get_vars()):
            v.set_elements_hash(1.0, 0.1, hash_id=k)
        return s

    # Every schedule option is a run-time toggle (plans are rebuilt, nothing is re-allocated), so by default ALL configurations of a
    # case are timed on ONE solution, i.e. on one set of var allocations, in --passes interleaved passes: where the arrays happen to
    # lie is worth 3-4 % of a step -- more than what some of these schedules differ by (profiles/r3_halves: the
    # "undivided" column of one case read 0.403 / 0.431 / 0.439 ms on three solutions).  --fresh-solutions restores one solution per
    # configuration.
    def own_solution(opts):
        return args.fresh_solutions

    for name, size, lo, hi in cases:
        shared = None
        for pass_no in range(max(1, args.passes)):
            for label, opts in configs:
                own = own_solution(opts)
                if own and pass_no > 0:
                    continue
                if own:
                    s = make(size, opts)
                else:
                    if shared is None:
                        shared = make(size, "")
                    s = shared
                    assert s.apply_command_line_options("-no-auto_tune " + opts) == ""
                    for k, v in enumerate(s.get_vars()):
                        v.set_elements_hash(1.0, 0.1, hash_id=k)
                ext, inter, whole = s.time_decomposed_step(lo, hi, reps=args.reps)
                pts = size[0] * size[1] * size[2]
                rec = {"case": name, "config": label, "shell_or_exterior_ms": round(ext, 4), "rest_or_interior_ms": round(inter, 4), "undivided_ms": round(whole, 4),
                       "overhead": round((ext + inter) / whole, 3), "shell_done_at_fraction": round(ext / (ext + inter), 3),
                       "gpoints_per_s_decomposed": round(pts / (ext + inter) * 1e-6, 1), "gpoints_per_s_undivided": round(pts / whole * 1e-6, 1),
                       "solution": "own" if own else "shared by the case's configurations", "pass": pass_no}
                out.append(rec)
                print(json.dumps(rec), flush=True)
                if own:
                    s.end_solution()
        if shared is not None:
            shared.end_solution()
    od = Path(__file__).resolve().parents[1] / "gpurun_out"
    od.mkdir(exist_ok=True)
    json.dump(out, open(od / f"decomp_cost_{args.stencil}.json", "w"), indent=1)


if __name__ == "__main__":
    main()
