#!/bin/bash
# GPU job r6u: 1-D solutions on the lifted vector point kernel: parity (one-tile and 2300-point goldens, every shape), then 2^24 points.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6u; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
( time timeout 1200 python3 -m pytest tests/test_reference_stencils_gpu.py -m gpu -q --timeout 300 -k "1d" 2>&1 | grep -v "^Solution '" ) > $O/tests.txt 2>&1
tail -n 12 $O/tests.txt
python3 tools/generic_table.py --out $O --only test_1d test_boundary_1d test_scratch_1d test_scratch_boundary_1d test_scratch_stages_1d test_stages_1d test_stream_1d test_func_1d test_step_cond_1d --tag one_d > $O/one_d.log 2>&1; cat $O/one_d.log
python3 - <<PY
import json
for r in json.load(open("$O/one_d.json")):
    print(r["stencil"], r["step_ms"], r["frac"], [(p["kernel"], p["ms"]) for p in r["parts"]][:6])
PY
