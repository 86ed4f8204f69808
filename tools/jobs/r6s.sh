#!/bin/bash
# GPU job r6s: 2-D solutions cut over 2 and 4 ranks against the reference fixtures (first test of its kind here).
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6s; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
( time timeout 1200 python3 -m pytest tests/test_two_d_ranks_gpu.py -m gpu -q --timeout 300 2>&1 | grep -v "^Solution '" ) > $O/tests.txt 2>&1
tail -n 40 $O/tests.txt
