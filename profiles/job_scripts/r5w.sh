#!/bin/bash
# GPU job r5w: the full GPU suite on the final tree of round 5 (with the multi-rank tests of the cluster shapes and the box lists).
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5w; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
timeout 1200 python3 -m pytest tests -m gpu -q --timeout 900 --durations=8 2>&1 | tail -30 > $O/gpu_tests.txt; tail -n 14 $O/gpu_tests.txt | cut -c1-220
timeout 120 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
