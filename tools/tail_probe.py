#!/usr/bin/env python
"""GPU box: what do the cheap tail planes (_tl shapes, ykh_starlin.hpp TAILOPT) buy at the x-chunk lengths that matter?
Times the named shapes over a box with forced x-chunks, and checks each _tl shape bit for bit against its plain sibling."""
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

SHAPES = {"iso3dfd": ["starlin_v4_z128_y32_r2_t2_nt_pd2_w2_c2", "starlin_v4_z128_y32_r2_t2_nt_pd2_tl_w2_c2", "starlin_v4_z128_y32_r2_t_nt_pd2_tl_w2_c2",
                      "starlin_v4_z128_y32_r2_m_nt_pd2_tl_w2_c2"],
          "3axis": ["starlin_v2_z64_y32_r2_u_nt_w2_c4", "starlin_v2_z64_y32_r2_u_nt_tl_w2_c4", "starlin_v2_z128_y32_r4_m_nt_w2_c4", "starlin_v2_z128_y32_r4_m_nt_tl_w2_c4"]}


def main():
    from yask_amd import yk_factory
    from yask_amd.kernel import yk_env
    yk_env.disable_debug_output()
    out = []
    quick = len(sys.argv) > 1
    for stencil, cases in (() if quick else (("iso3dfd", [((512, 512, 512), (0, 64)), ((1024, 1024, 512), (0, 256)), ((1024, 1024, 1024), (0,))]),
                           ("3axis", [((512, 512, 512), (0, 64)), ((1024, 1024, 1024), (0,))]))):
        fac = yk_factory(stencil)
        for size, chunks in cases:
            for name in SHAPES[stencil]:
                s = fac.new_solution(fac.new_env())
                s.set_overall_domain_size_vec(list(size))
                assert s.apply_command_line_options("-no-auto_tune") == ""
                s.prepare_solution()
                names = s.get_kernel_variant_names(0)
                if name not in names:
                    print("missing", name); s.end_solution(); continue
                vi = names.index(name)
                for k, v in enumerate(s.get_vars()):
                    v.set_elements_hash(1.0, 0.1, hash_id=k)
                for xc in chunks:
                    s.time_part(0, vi, xc, 0, 3)
                    ms = s.time_part(0, vi, xc, 0, 20)
                    rec = {"stencil": stencil, "size": size, "variant": name, "xchunk": xc, "ms": round(ms, 4), "gpoints_per_s": round(size[0] * size[1] * size[2] / ms * 1e-6, 1)}
                    out.append(rec)
                    print(json.dumps(rec), flush=True)
                s.end_solution()
    # bit-exactness of the _tl shapes against their plain siblings (ragged size, forced short chunks, 3 steps)
    for stencil, a, b in (("iso3dfd", SHAPES["iso3dfd"][0], SHAPES["iso3dfd"][0]), ("iso3dfd", SHAPES["iso3dfd"][0], "starlin_v4_z128_y32_r2_t_nt_pd2_w2_c2"),
                          ("iso3dfd", SHAPES["iso3dfd"][0], SHAPES["iso3dfd"][1]), ("iso3dfd", SHAPES["iso3dfd"][0], SHAPES["iso3dfd"][2]),
                          ("3axis", SHAPES["3axis"][0], SHAPES["3axis"][1]), ("3axis", SHAPES["3axis"][2], SHAPES["3axis"][3])):
        fac = yk_factory(stencil)
        res = []
        for name in (a, b):
            s = fac.new_solution(fac.new_env())
            s.set_overall_domain_size_vec([150, 77, 200])
            assert s.apply_command_line_options(f"-no-auto_tune -hip_variant {name} -hip_xchunk 37") == ""
            s.prepare_solution()
            for k, v in enumerate(s.get_vars()):
                v.set_elements_hash(1.0, 0.1, hash_id=k)
            s.run_solution(0, 2)
            vn = s.get_vars()[0].get_name()
            res.append(s.get_var(vn).get_elements_in_slice([3, 0, 0, 0], [3, 149, 76, 199])[0].copy())
            s.end_solution()
        d = res[0] != res[1]
        xs = np.nonzero(d.any(axis=(1, 2)))[0]
        print("bit-identical", stencil, b, "vs", a, ":", bool(np.array_equal(res[0], res[1])), "differing points", int(d.sum()), "max |diff|",
              float(np.abs(res[0].astype(np.float64) - res[1]).max()), "x planes", xs[:12].tolist(), "...", xs[-4:].tolist(), flush=True)
    json.dump(out, open(Path(__file__).resolve().parents[1] / "gpurun_out" / "tail_probe.json", "w"), indent=1)


if __name__ == "__main__":
    main()
