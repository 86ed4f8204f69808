#!/bin/bash
# GPU job r6x: fused scratch groups with several tile shapes timed by prepare_solution(): parity, choices, times.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6x; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
( timeout 900 python3 -m pytest tests/test_fused_scratch_gpu.py tests/test_two_d_ranks_gpu.py tests/test_reference_api_programs_gpu.py -m gpu -q --timeout 300 2>&1 | grep -v "^Solution '" ) > $O/tests.txt 2>&1
tail -n 5 $O/tests.txt
python3 tools/generic_table.py --out $O --only swe2d wave2d wave2d_f64 test_scratch_2d --tag default > $O/default.log 2>&1; cat $O/default.log
python3 - <<PY
import sys
sys.path.insert(0, "$R")
from yask_amd import yk_factory
from yask_amd.kernel import yk_env
yk_env.disable_debug_output()
for st in ("wave2d", "swe2d", "wave2d_f64", "test_scratch_2d"):
    for n in (1024, 4096):
        fac = yk_factory(st)
        s = fac.new_solution(fac.new_env())
        s.set_overall_domain_size_vec([n, n])
        s.prepare_solution()
        print(st, n, "fused groups:", s.get_fused_groups())
        s.end_solution()
PY
