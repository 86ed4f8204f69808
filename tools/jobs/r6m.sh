#!/bin/bash
# GPU job r6m: 8-byte-lane vector point kernel for parts with many operands: parity + every shape of test_partial_3d / cube / tti timed.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6m; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
( timeout 1200 python3 -m pytest tests/test_multi_tile_fixtures_gpu.py tests/test_reference_stencils_gpu.py tests/test_box_kernel_gpu.py -m gpu -q --timeout 600 -k "partial or cube or 3plane or diags or tti or round_5" 2>&1 | grep -v "^Solution '" ) > $O/tests.txt 2>&1
tail -n 4 $O/tests.txt
python3 - <<PY
import sys
sys.path.insert(0, "$R")
from yask_amd import yk_factory
from yask_amd.kernel import yk_env
yk_env.disable_debug_output()
for st in ("test_partial_3d", "tti"):
    fac = yk_factory(st)
    s = fac.new_solution(fac.new_env())
    s.set_overall_domain_size_vec([512, 512, 512])
    s.prepare_solution()
    for k, v in enumerate(s.get_vars()):
        v.set_elements_hash(1.0 + 0.25 * k, 0.1, hash_id=k)
    print(st, "chosen:", s.get_kernel_variant(0))
    for i, n in enumerate(s.get_kernel_variant_names(0)):
        if "vecpt" in n or n == "naive" or n == s.get_kernel_variant(0):
            s.time_part(part=0, variant=i, t=0, reps=1)
            print("  %-44s %.3f ms  scratch %d B" % (n, s.time_part(part=0, variant=i, t=0, reps=5), s.get_kernel_variant_scratch_bytes(0, i)))
    s.end_solution()
PY
