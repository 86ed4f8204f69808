#!/bin/bash
# GPU job r6e: fused scratch groups (csrc/ykh_fused.hpp) -- parity against the reference goldens, then swe2d / wave2d step times.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6e; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
( time timeout 900 python3 -m pytest tests/test_fused_scratch_gpu.py -m gpu -q --timeout 300 2>&1 | grep -v "^Solution '" ) > $O/fused_tests.txt 2>&1
tail -n 30 $O/fused_tests.txt
( time timeout 900 python3 -m pytest tests/test_reference_stencils_gpu.py tests/test_reference_api_programs_gpu.py tests/test_step_graphs_gpu.py -m gpu -q --timeout 300 -x 2>&1 | grep -v "^Solution '" ) > $O/other_tests.txt 2>&1
tail -n 6 $O/other_tests.txt
TWO="swe2d wave2d wave2d_f64 test_scratch_2d"
YASK_HIP_FUSE_SCRATCH=0 python3 tools/generic_table.py --out $O --only $TWO --tag unfused > $O/unfused.log 2>&1; cat $O/unfused.log
python3 tools/generic_table.py --out $O --only $TWO --tag fused > $O/fused.log 2>&1; cat $O/fused.log
python3 - <<PY
import sys
sys.path.insert(0, "$R")
from yask_amd import yk_factory
from yask_amd.kernel import yk_env
yk_env.disable_debug_output()
for st in ("wave2d", "swe2d"):
    for n in (1024, 4096, 8192):
        fac = yk_factory(st)
        s = fac.new_solution(fac.new_env())
        s.set_overall_domain_size_vec([n, n])
        s.apply_command_line_options("-hip_step_timers")
        s.prepare_solution()
        print(st, n, "fused groups:", s.get_fused_groups())
        s.end_solution()
PY
