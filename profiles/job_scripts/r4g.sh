#!/bin/bash
# GPU job r4g: after ipc_free stopped being a collective: the dry-run bench tests (N = 2, 4, 8 on one GPU over gloo, everything on auto)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4g; mkdir -p $O; cd $R
export YASK_BENCH_STACK_DUMP_S=100
YASK_TEST_MULTI_DEVICE_DRYRUN=1 timeout 420 python -m pytest tests/test_multi_device_gpu.py -m gpu -q -x -k "bench_on_real" --durations=5 2>&1 | tail -60 > $O/dryrun_bench.log; tail -30 $O/dryrun_bench.log | cut -c1-400
