#!/usr/bin/env python
"""GPU box: one REGULAR launch of the default kernel over a box, with forced x-chunk lengths -- what do two (or more) rounds of
equal blocks cost against the one-round default?  (The planned launches of a decomposed rank are rounds of equal blocks.)"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    from yask_amd import yk_factory
    from yask_amd.kernel import yk_env
    yk_env.disable_debug_output()
    out = []
    for stencil, size, chunks in (("iso3dfd", (512, 512, 512), (0, 128, 64, 43, 32)), ("iso3dfd", (1024, 1024, 512), (0, 512, 256, 171, 128)),
                                  ("iso3dfd", (1024, 1024, 1024), (0, 512, 256)), ("ssg", (512, 512, 512), (0, 256, 128, 86, 64))):
        fac = yk_factory(stencil)
        for xc in chunks:
            s = fac.new_solution(fac.new_env())
            s.set_overall_domain_size_vec(list(size))
            assert s.apply_command_line_options(f"-no-auto_tune -hip_xchunk {xc}") == ""
            s.prepare_solution()
            for k, v in enumerate(s.get_vars()):
                v.set_elements_hash(1.0, 0.1, hash_id=k)
            ms = sum(s.time_part(part=p, variant=-1, t=0, reps=20) for p in range(s.get_num_parts()))
            rec = {"stencil": stencil, "size": size, "xchunk": xc, "ms": round(ms, 4), "gpoints_per_s": round(size[0] * size[1] * size[2] / ms * 1e-6, 1)}
            out.append(rec)
            print(json.dumps(rec), flush=True)
            s.end_solution()
    od = Path(__file__).resolve().parents[1] / "gpurun_out"
    od.mkdir(exist_ok=True)
    json.dump(out, open(od / "xchunk_probe.json", "w"), indent=1)


if __name__ == "__main__":
    main()
