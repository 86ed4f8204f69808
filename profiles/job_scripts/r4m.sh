#!/bin/bash
# GPU job r4m: the tree as it will be handed over -- whole GPU suite, the multi-device file in dry-run mode, smoke(), the default bench line
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4m; mkdir -p $O; cd $R
timeout 600 python -m pytest tests -m gpu -q --durations=6 > $O/suite.log 2>&1; grep -E "passed|failed" $O/suite.log | tail -2; grep -E "^FAILED|^ERROR" $O/suite.log | head
YASK_TEST_MULTI_DEVICE_DRYRUN=1 timeout 400 python -m pytest tests/test_multi_device_gpu.py -m gpu -q --durations=4 > $O/multidev_dryrun.log 2>&1; grep -E "passed|failed" $O/multidev_dryrun.log | tail -2; grep -E "^FAILED|^ERROR" $O/multidev_dryrun.log | head
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; j=json.load(open('$O/bench_default.json')); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['traffic'], j['bandwidth_probe'], {k: j['cpu_baseline'][k] for k in ('value','no_tune_best','auto_tuned')})"
