#!/bin/bash
# GPU job r6h: the whole GPU suite on the round-6 tree (as the driver runs it: -x -q -m gpu) + smoke + default bench.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6h; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
( time timeout 1700 python3 -m pytest tests -m gpu -x -q --timeout 600 --durations=15 2>&1 | grep -v "^Solution '" ) > $O/gpu_tests.txt 2>&1
tail -n 30 $O/gpu_tests.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^Solution '" | tail -2
timeout 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; python3 -c "
import json; j=json.loads([l for l in open('$O/bench_default.json') if l.startswith('{')][0]); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['traffic'], j['cpu_baseline']['value'])"
