#!/bin/bash
# GPU job r6r: plane-ring kernel with the saddr form for its where-used (kind 3) reads: parity, then tti & co at 512^3.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6r; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
( timeout 1200 python3 -m pytest tests/test_multi_tile_fixtures_gpu.py tests/test_reference_stencils_gpu.py tests/test_box_kernel_gpu.py tests/test_fused_scratch_gpu.py -m gpu -q --timeout 600 2>&1 | grep -v "^Solution '" ) > $O/tests.txt 2>&1
tail -n 4 $O/tests.txt
python3 tools/generic_table.py --out $O --only tti cube 3plane 3axis_with_diags test_scratch_3d awp_abc test_partial_3d --size3 512 --tag after > $O/after.log 2>&1; cat $O/after.log
python3 - <<PY
import json
for r in json.load(open("$O/after.json")):
    print(r["stencil"], r["step_ms"], r["frac"], [(p["kernel"], p["ms"]) for p in r["parts"]][:7])
PY
