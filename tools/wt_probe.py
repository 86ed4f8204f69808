#!/usr/bin/env python
"""GPU box: do WRITE-THROUGH (sc1) output stores help?  (round 4; ykh_starlin.hpp stv_b_wt)

MI355X_MICROARCH.md: plain / sc0 / nt stores keep the written line in the XCD's L2, sc1 stores drop it.  The marching kernels' output
stream is never read again within a sweep; with `nt` it still occupies the L2 next to the halo lines of the arriving planes
(3axis fp64: 5 % of the reads are halo lines that fell out of the L2, profiles/r4_3axis_fetch).  The `_wt` shapes differ from their
siblings in that one instruction.  ONE solution per case (one set of allocations), the two shapes timed alternately, `passes` times;
then both shapes on a ragged grid, bit for bit.

    python tools/wt_probe.py [passes]"""
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

PAIRS = {  # stencil -> [(part, plain shape, write-through shape)]
    "iso3dfd": [(0, "starlin_v4_z128_y32_r2_t2_nt_pd2_tl_w2_c2", "starlin_v4_z128_y32_r2_t2_nt_pd2_tl_wt_w2_c2")],
    "3axis": [(0, "starlin_v2_z128_y32_r4_m_nt_w2_c4", "starlin_v2_z128_y32_r4_m_nt_wt_w2_c4"),
              (0, "starlin_v2_z64_y32_r2_u_nt_tl_w2_c4", "starlin_v2_z64_y32_r2_u_nt_tl_wt_w2_c4")],
    "ssg": [(0, "march_v4_z128_y16_nt_hr_ps_fd_t2_w2", "march_v4_z128_y16_nt_hr_ps_fd_t2_wt_w2"),
            (1, "march_v4_z128_y16_nt_hr_fd_w2", "march_v4_z128_y16_nt_hr_fd_wt_w2")],
}
CASES = [("iso3dfd", 1024, [0]), ("iso3dfd", 512, [0]), ("3axis", 1024, [0]), ("3axis", 512, [1]), ("ssg", 512, [0, 1])]


def main():
    from yask_amd import yk_factory
    from yask_amd.kernel import yk_env
    yk_env.disable_debug_output()
    passes = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    out = []
    for stencil, n, which in CASES:
        fac = yk_factory(stencil)
        s = fac.new_solution(fac.new_env())
        s.set_overall_domain_size_vec([n, n, n])
        assert s.apply_command_line_options("-no-auto_tune") == ""
        s.prepare_solution()
        for k, v in enumerate(s.get_vars()):
            v.set_elements_hash(1.0 + 0.25 * k, 0.1, hash_id=k)
        for w in which:
            part, a, b = PAIRS[stencil][w]
            names = s.get_kernel_variant_names(part)
            ia, ib = names.index(a), names.index(b)
            reps = 30 if n <= 512 else 12
            for vi in (ia, ib):
                s.time_part(part, vi, 0, 0, 3)
            ms = {a: [], b: []}
            for _ in range(passes):
                for name, vi in ((a, ia), (b, ib)):
                    ms[name].append(s.time_part(part, vi, 0, 0, reps))
            rec = {"stencil": stencil, "size": n, "part": part, "plain": a, "plain_ms": [round(x, 4) for x in ms[a]],
                   "wt": b, "wt_ms": [round(x, 4) for x in ms[b]], "wt_over_plain": round(min(ms[b]) / min(ms[a]), 4)}
            out.append(rec)
            print(json.dumps(rec), flush=True)
        s.end_solution()
    # same bits?  ragged grid, three steps, forced short x-chunks
    for stencil, pairs in PAIRS.items():
        fac = yk_factory(stencil)
        for part, a, b in pairs:
            res = []
            for name in (a, b):
                s = fac.new_solution(fac.new_env())
                s.set_overall_domain_size_vec([150, 77, 200])
                # (ssg: a name exists in ONE part only, the other part keeps its default)
                rem = s.apply_command_line_options(f"-no-auto_tune -hip_xchunk 37 -hip_variant {name}")
                if rem != "":
                    print("cannot name the shape of one part:", rem)
                    res = None
                    break
                s.prepare_solution()
                for k, v in enumerate(s.get_vars()):
                    v.set_elements_hash(1.0 + 0.25 * k, 0.1, hash_id=k)
                s.run_solution(0, 2)
                got = []
                for v in s.get_vars():
                    if v.get_num_dims() == 4:
                        t = v.get_last_valid_step_index()
                        got.append(v.get_elements_in_slice([t, 0, 0, 0], [t, 149, 76, 199])[0].copy())
                res.append(got)
                s.end_solution()
            if res:
                same = all(np.array_equal(x, y) for x, y in zip(res[0], res[1]))
                print(json.dumps({"stencil": stencil, "part": part, "wt_bit_identical_to_plain": bool(same)}), flush=True)
                out.append({"stencil": stencil, "part": part, "wt_bit_identical_to_plain": bool(same)})
    p = Path(__file__).resolve().parents[1] / "gpurun_out" / "wt_probe.json"
    p.parent.mkdir(exist_ok=True)
    json.dump(out, open(p, "w"), indent=1)


if __name__ == "__main__":
    main()
