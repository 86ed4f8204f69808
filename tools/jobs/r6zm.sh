#!/bin/bash
# GPU job r6zm: plane-ring shapes for small parts that are mostly mixed-offset reads (test_3d / test_stages_3d / test_boundary_3d, the 2-D
# test stencils): the table, then the whole GPU suite (every registered shape of every solution against the reference fixtures).
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6zm; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
python3 tools/generic_table.py --out $O --only test_3d test_stages_3d test_boundary_3d test_2d test_boundary_2d test_reverse_2d test_scratch_2d test_3d-zyx test_stages_3d-xzy test_2d-yx test_reverse_2d-r1 --size3 512 --tag box > $O/box.log 2>&1; cat $O/box.log
( time timeout 1700 python3 -m pytest tests -m gpu -x -q --timeout 600 2>&1 | grep -v "^Solution '" ) > $O/gpu_tests.txt 2>&1
grep -n "passed\|failed" $O/gpu_tests.txt | tail -3
