#!/bin/bash
# GPU job r6f: fused scratch groups after the per-part boxes: parity, then fused (forced) vs unfused (forced) step times.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6f; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
( time timeout 900 python3 -m pytest tests/test_fused_scratch_gpu.py -m gpu -q --timeout 300 2>&1 | grep -v "^Solution '" ) > $O/fused_tests.txt 2>&1
tail -n 8 $O/fused_tests.txt
TWO="swe2d wave2d wave2d_f64 test_scratch_2d"
YASK_HIP_FUSE_SCRATCH=0 python3 tools/generic_table.py --out $O --only $TWO --tag unfused > $O/unfused.log 2>&1; cat $O/unfused.log
YASK_HIP_FUSE_SCRATCH=1 python3 tools/generic_table.py --out $O --only $TWO --tag fused > $O/fused.log 2>&1; cat $O/fused.log
