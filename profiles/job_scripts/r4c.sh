#!/bin/bash
# GPU job r4c: IPC control-plane counters; the multi-device test file dry-run on one GPU (IPC + bench.py self-check over gloo);
# bench.py's N>1 flow tests (self-check added)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4c; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_transport_gpu.py -m gpu -x -q -k "keeps_the_host or one_rank_only or native_bootstrap" 2>&1 | tail -40 > $O/transport.log; tail -5 $O/transport.log
YASK_TEST_MULTI_DEVICE_DRYRUN=1 timeout 1200 python -m pytest tests/test_multi_device_gpu.py -m gpu -q --durations=8 2>&1 | tail -80 > $O/multidev_dryrun.log; tail -30 $O/multidev_dryrun.log
timeout 600 python -m pytest tests/test_multirank_gpu.py -m gpu -x -q -k bench 2>&1 | tail -40 > $O/bench_tests.log; tail -8 $O/bench_tests.log
