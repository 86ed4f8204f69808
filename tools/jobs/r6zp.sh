#!/bin/bash
# GPU job r6zp: the end of round 6: the whole GPU suite, smoke, the default bench line on the final tree.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6zp; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
( time timeout 1700 python3 -m pytest tests -m gpu -x -q --timeout 600 2>&1 | grep -v "^Solution '" ) > $O/gpu_tests.txt 2>&1
grep -n "passed\|failed" $O/gpu_tests.txt | tail -3
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^Solution '" | tail -1
timeout 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; python3 -c "
import json; j=json.loads([l for l in open('$O/bench_default.json') if l.startswith('{')][0]); print('bench', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['traffic'], j['cpu_baseline']['value'])"
timeout 1500 python3 tools/generic_table.py --out $O --size3 512 --tag final > $O/table_final.log 2>&1; tail -n 70 $O/table_final.log
