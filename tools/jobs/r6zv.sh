#!/bin/bash
# GPU job r6zv: smoke and the default bench line on the final tree of round 6 (the GPU suite of this tree: job r6zu).
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6zv; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^Solution '" | tail -1
timeout 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; python3 -c "
import json; j=json.loads([l for l in open('$O/bench_default.json') if l.startswith('{')][0]); print('bench', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['traffic'], j['cpu_baseline']['value'])"
python3 tools/generic_table.py --out $O --only awp awp_elastic ssg2 cube tti iso3dfd_sponge test_3d --size3 512 --tag last 2>&1
