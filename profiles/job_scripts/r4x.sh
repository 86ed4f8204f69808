#!/bin/bash
# GPU job r4x (last of the round): the remaining IPC users after the packed-x change: config 4's full grid over 8 ranks, the multirank file, the multi-device file in dry-run mode
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4x; mkdir -p $O; cd $R
timeout 60 python -m pytest tests/test_big_fixtures_gpu.py -m gpu -q -k "config4" > $O/big.log 2>&1; grep -E "passed|failed" $O/big.log | tail -1
YASK_TEST_MULTI_DEVICE_DRYRUN=1 timeout 100 python -m pytest tests/test_multi_device_gpu.py -m gpu -q -x > $O/dry.log 2>&1; grep -E "passed|failed" $O/dry.log | tail -1
timeout 60 python -m pytest tests/test_multirank_gpu.py -m gpu -q -x -k "not bench" > $O/multirank.log 2>&1; grep -E "passed|failed" $O/multirank.log | tail -1
