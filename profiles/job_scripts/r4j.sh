#!/bin/bash
# GPU job r4j: numbers of the 20-step ssg pin (printed by the test), transport / multirank / multi-device dry-run after the waiter early-exit
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4j; mkdir -p $O; cd $R
timeout 120 python -m pytest tests/test_big_fixtures_gpu.py -m gpu -q -s -k twenty 2>&1 | grep -E "ssg 256|exact-division|passed|failed" | cut -c1-900 | tee $O/ssg20.txt
timeout 300 python -m pytest tests/test_transport_gpu.py tests/test_multirank_gpu.py -m gpu -q -x 2>&1 | tail -4
YASK_TEST_MULTI_DEVICE_DRYRUN=1 timeout 200 python -m pytest tests/test_multi_device_gpu.py -m gpu -q -x -k "bench_on_real or two_devices" 2>&1 | tail -4
