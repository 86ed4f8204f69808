#!/bin/bash
# GPU job r6zr: what does the slab schedule (exterior slabs, then the interior) cost the generic solutions on the compute side?
# Corner rank of a 2 x 2 x 2 grid (one neighbour per dim), local 256^3 and 512^3, against the undivided box; no communication.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6zr; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
python3 - <<'PY' | tee $O/slab_cost.txt
import sys, json, time
sys.path.insert(0, '.')
from yask_amd import yk_factory
from yask_amd.kernel import yk_env
yk_env.disable_debug_output()
ramped = False
for stencil in ["awp", "awp_elastic", "ssg2", "cube", "iso3dfd_sponge", "tti", "fsg"]:
    for n in (256, 512):
        if stencil == "fsg" and n == 512: continue
        fac = yk_factory(stencil)
        s = fac.new_solution(fac.new_env())
        s.set_overall_domain_size_vec([n, n, n])
        s.prepare_solution()
        if not ramped:
            t0, t = time.perf_counter(), 0
            while time.perf_counter() - t0 < 1.5:
                s.run_solution(t, t + 9); t += 10
            ramped = True
        for k, v in enumerate(s.get_vars()):
            v.set_elements_hash(1.0 + 0.25 * k, 0.1, hash_id=k)
        best = None
        for rep in range(2):
            ext, inter, whole = s.time_decomposed_step((0, 0, 0), (1, 1, 1), reps=6)
            r = (ext + inter) / whole
            if best is None or r < best[0]: best = (r, ext, inter, whole)
        print(json.dumps({"stencil": stencil, "local": n, "exterior_ms": round(best[1], 4), "interior_ms": round(best[2], 4), "undivided_ms": round(best[3], 4), "overhead": round(best[0], 3)}), flush=True)
PY
