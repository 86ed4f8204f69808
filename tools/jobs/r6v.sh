#!/bin/bash
# GPU job r6v: fused scratch kernel on awkward grid sizes; then the whole suite once more on the final tree.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6v; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
( time timeout 900 python3 -m pytest tests/test_fused_scratch_gpu.py -m gpu -q --timeout 300 2>&1 | grep -v "^Solution '" ) > $O/fused.txt 2>&1
tail -n 25 $O/fused.txt
( time timeout 1700 python3 -m pytest tests -m gpu -x -q --timeout 600 2>&1 | grep -v "^Solution '" ) > $O/gpu_tests.txt 2>&1
tail -n 6 $O/gpu_tests.txt
