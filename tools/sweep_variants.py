#!/usr/bin/env python
"""GPU-side sweep: time every compiled kernel variant (x several x-chunk lengths) with HIP events and
check each against the generic naive kernel at full size. Writes gpurun_out/sweep_<stencil>_<n>.json."""
import argparse
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stencil", default="iso3dfd")
    ap.add_argument("--size", type=int, nargs="+", default=[1024])
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--chunks", type=int, nargs="+", default=[0, 1024, 256, 128, 64])
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--bytes-per-point", type=float, default=16.0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--part", type=int, default=0)
    ap.add_argument("--steps", type=int, default=0, help="also time N whole steps with the default variants")
    args = ap.parse_args()
    from yask_amd import yk_factory
    size = args.size if len(args.size) == 3 else [args.size[0]] * 3
    fac = yk_factory(args.stencil)
    env = fac.new_env()

    def init(s):
        # var k = 1 + k/4 + 0.1*hash(logical index): O(1) data, the same in every solution (as yask_amd/harness.py)
        for k, v in enumerate(s.get_vars()):
            v.set_elements_hash(1.0 + 0.25 * k, 0.1, hash_id=k)

    def make():
        s = fac.new_solution(env)
        s.set_overall_domain_size_vec(size)
        s.prepare_solution()
        init(s)
        return s

    soln = make()
    ref = None
    if args.check:
        ref = make()
        ref.apply_command_line_options("-force_scalar")
        ref.prepare_solution()
        init(ref)
        ref.run_solution(0, 1)
    pts = float(size[0]) * size[1] * size[2]
    names = soln.get_kernel_variant_names(args.part)
    results = []
    for vi, name in enumerate(names):
        chunks = args.chunks if name != "naive" else [0]
        for xc in chunks:
            try:
                soln.time_part(args.part, vi, xc, 0, 2)
                ms = soln.time_part(args.part, vi, xc, 0, args.reps)
            except RuntimeError as e:
                print("FAILED", name, xc, e, flush=True)
                continue
            gp = pts / (ms * 1e-3) * 1e-9
            rec = {"variant": name, "xchunk": xc, "ms": round(ms, 4), "gpoints_per_s": round(gp, 2),
                   "gbs_algorithmic": round(gp * args.bytes_per_point, 1)}
            results.append(rec)
            print(rec, flush=True)
        if args.check and name != "naive" and not name.startswith("abl"):
            chk = make()
            chk.apply_command_line_options(f"-hip_variant {name}")
            chk.prepare_solution()
            init(chk)
            chk.run_solution(0, 1)
            bad = chk.compare_data(ref, 1e-4)
            print("check", name, "mismatches vs naive:", bad, flush=True)
            results.append({"variant": name, "mismatches_vs_naive": bad})
            chk.end_solution()
    if args.steps > 0:
        soln.run_solution(0, 1)
        soln.get_stats()
        soln.run_solution(2, 1 + args.steps)
        st = soln.get_stats()
        rec = {"whole_step_ms": round(st.get_elapsed_secs() / args.steps * 1e3, 4),
               "gpoints_per_s": round(pts * args.steps / st.get_elapsed_secs() * 1e-9, 2),
               "variants": [soln.get_kernel_variant(i) for i in range(8) if i < soln.get_num_parts()]}
        results.append(rec)
        print("WHOLE STEP:", rec, flush=True)
    results_sorted = sorted([r for r in results if "ms" in r], key=lambda r: r["ms"])
    out = args.out or str(ROOT / "gpurun_out" / f"sweep_{args.stencil}_p{args.part}_{size[0]}x{size[1]}x{size[2]}.json")
    Path(out).parent.mkdir(parents=True, exist_ok=True)
    json.dump({"size": size, "results": results, "best": results_sorted[:5]}, open(out, "w"), indent=1)
    print("BEST:", results_sorted[:5])


if __name__ == "__main__":
    main()
