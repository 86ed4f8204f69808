#!/bin/bash
# GPU job r6zt: the whole GPU suite on the tree with descriptor-reading twins for the generic marching shapes.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6zt; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
( time timeout 1700 python3 -m pytest tests -m gpu -x -q --timeout 600 2>&1 | grep -v "^Solution '" ) > $O/gpu_tests.txt 2>&1
grep -n "passed\|failed\|Error" $O/gpu_tests.txt | tail -5
