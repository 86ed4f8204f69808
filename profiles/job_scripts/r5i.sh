#!/bin/bash
# GPU job r5i: the full GPU suite with per-test durations (r5h's run took 701 s where r5b / r5e took 185 s: the box, or a change?)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5i; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
uptime > $O/host.txt; nproc >> $O/host.txt
timeout 1500 python3 -m pytest tests -m gpu -q --timeout 900 --durations=40 2>&1 | tail -70 > $O/gpu_tests.txt; tail -50 $O/gpu_tests.txt
uptime >> $O/host.txt; cat $O/host.txt
