#!/usr/bin/env python
"""GPU box: does a soft LOCK-STEP of the workgroups that share an XCD stop the 3axis fp64 kernel from re-fetching halo lines?
(round 5; the "_ls<K>" shapes of ykh_starlin.hpp, profiling build: YASK_HIP_LIB_DIR=yask_amd/lib_prof)

profiles/r4_3axis_fetch: at 1024^3, 5 % of the reads of the default large-grid shape are halo lines of a tile that its neighbour
streamed as interior but that fell out of the XCD's 4 MiB L2 before the tile asked -- the 32 workgroups of an XCD drift planes
apart.  The _ls<K> shapes make wave 0 of every workgroup count itself in at a per-XCD counter every K planes and wait (bounded)
for the XCD's other workgroups.  ONE solution, one set of allocations, the shapes timed alternately `passes` times; with
--fetch <shape> a single shape runs 6 launches (for a rocprofv3 --pmc pass around this script); bit-identity on a ragged grid.

    python tools/lockstep_probe.py [--passes 3] [--size 1024] | --fetch SHAPE"""
import argparse
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
BASES = {"3axis": "starlin_v2_z128_y32_r4_m_nt_w2_c4", "3axis512": "starlin_v2_z64_y32_r2_u_nt_tl_w2_c4", "iso3dfd": "starlin_v4_z128_y32_r2_t2_nt_pd2_tl_w2_c2"}
FIELD = {"3axis": "A", "3axis512": "A", "iso3dfd": "p"}


def ls_names(base):
    return [base.replace("_w2_", f"_ls{k}_w2_") for k in (1, 2, 4, 8, 16, 32, 64)]


def make(fac, n, opts="-no-auto_tune"):
    s = fac.new_solution(fac.new_env())
    s.set_overall_domain_size_vec([n, n, n] if isinstance(n, int) else n)
    assert s.apply_command_line_options(opts) == ""
    s.prepare_solution()
    for k, v in enumerate(s.get_vars()):
        v.set_elements_hash(1.0 + 0.25 * k, 0.1, hash_id=k)
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--fetch", default=None)
    ap.add_argument("--stencil", default="3axis", choices=sorted(BASES) + ["ssg"])
    ap.add_argument("--part", type=int, default=0)
    ap.add_argument("--shapes", nargs="*", default=None, help="explicit shape names, the first one is the base (default: the stencil's base + its _ls shapes)")
    ap.add_argument("--only", type=int, nargs="*", default=None, help="lock-step periods to time (default: all compiled)")
    ap.add_argument("--no-bits", action="store_true")
    args = ap.parse_args()
    from yask_amd import yk_factory
    from yask_amd.kernel import yk_env
    yk_env.disable_debug_output()
    fac = yk_factory(args.stencil.replace("512", ""))
    part = args.part
    s = make(fac, args.size)
    names = s.get_kernel_variant_names(part)
    if args.fetch:
        s.time_part(part, names.index(args.fetch), 0, 0, 6)
        return
    if args.shapes:
        BASE, shapes = args.shapes[0], [x for x in args.shapes if x in names]
        missing = [x for x in args.shapes if x not in names]
        if missing:
            print("not in this library:", missing, flush=True)
    else:
        BASE, LS = BASES[args.stencil], ls_names(BASES[args.stencil])
        shapes = [BASE] + [x for x in LS if x in names and (not args.only or int(x.split("_ls")[1].split("_")[0]) in args.only)]
    if len(shapes) < 2:
        raise SystemExit("no _ls shapes in this library: build one with tools/build_prof_lib.sh <stencil> and set YASK_HIP_LIB_DIR")
    idx = {x: names.index(x) for x in shapes}
    reps = 12 if args.size >= 1024 else 30
    for x in shapes:
        s.time_part(part, idx[x], 0, 0, 3)
    ms = {x: [] for x in shapes}
    for _ in range(args.passes):
        for x in shapes:
            ms[x].append(s.time_part(part, idx[x], 0, 0, reps))
    s.end_solution()
    out = {"stencil": args.stencil, "part": part, "size": args.size, "ms": {x: [round(v, 4) for v in ms[x]] for x in shapes},
           "best_over_base": {x: round(min(ms[x]) / min(ms[BASE]), 4) for x in shapes}}
    print(json.dumps(out), flush=True)
    # same bits?  a grid whose tile count is a multiple of 8 (the lock-step is live) and a ragged one (it is not)
    same = {}
    for size in ([] if args.no_bits else ([96, 256, 512], [150, 77, 200])):
        res = []
        for x in shapes:
            q = make(fac, size, f"-no-auto_tune -hip_variant {x}")
            q.run_solution(0, 2)
            got = []
            for v in q.get_vars():
                if v.get_num_dims() == 4:
                    t = v.get_last_valid_step_index()
                    got.append(v.get_elements_in_slice([t, 0, 0, 0], [t, size[0] - 1, size[1] - 1, size[2] - 1])[0].copy())
            res.append(got)
            q.end_solution()
        same["x".join(map(str, size))] = all(all(np.array_equal(a, b) for a, b in zip(res[0], r)) for r in res[1:])
    out["bit_identical_to_base"] = same
    print(json.dumps(same), flush=True)
    p = Path(__file__).resolve().parents[1] / "gpurun_out" / "r5_lockstep"
    p.mkdir(parents=True, exist_ok=True)
    json.dump(out, open(p / f"lockstep_probe_{args.stencil}_p{part}_{args.size}.json", "w"), indent=1)


if __name__ == "__main__":
    main()
