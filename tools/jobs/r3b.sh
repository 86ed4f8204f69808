#!/bin/bash
# GPU job r3b: where does a decomposed run differ from one rank (tools/diag_bitexact.py)?  rounds-of-equal-blocks planner vs the
# first planner vs slabs (tools/decomp_cost.py); regular launches with forced x-chunks (tools/xchunk_probe.py); the tests that failed in r3a.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3b
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== bit-exactness diagnostic"; ( time timeout 600 python tools/diag_bitexact.py ) > $O/diag.log 2>&1; echo "rc=$?"; tail -60 $O/diag.log
echo "== xchunk probe"; ( time timeout 300 python tools/xchunk_probe.py ) > $O/xchunk.log 2>&1; echo "rc=$?"; cat $O/xchunk.log | tail -22
echo "== decomp cost iso3dfd"; ( time timeout 420 python tools/decomp_cost.py --stencil iso3dfd ) > $O/decomp_iso3dfd.log 2>&1; echo "rc=$?"; cp gpurun_out/decomp_cost_iso3dfd.json $O/ 2>/dev/null; python - <<'PY'
import json
for r in json.load(open("gpurun_out/decomp_cost_iso3dfd.json")):
    print(r["case"][:34].ljust(34), r["config"][:34].ljust(34), r["shell_or_exterior_ms"], r["rest_or_interior_ms"], r["undivided_ms"], r["overhead"], r["shell_done_at_fraction"])
PY
echo "== decomp cost ssg"; ( time timeout 300 python tools/decomp_cost.py --stencil ssg ) > $O/decomp_ssg.log 2>&1; echo "rc=$?"; cp gpurun_out/decomp_cost_ssg.json $O/ 2>/dev/null; python - <<'PY'
import json
for r in json.load(open("gpurun_out/decomp_cost_ssg.json")):
    print(r["case"][:34].ljust(34), r["config"][:34].ljust(34), r["shell_or_exterior_ms"], r["rest_or_interior_ms"], r["undivided_ms"], r["overhead"], r["shell_done_at_fraction"])
PY
echo "== tests"; ( time timeout 600 python -m pytest tests/test_fuse_vars_gpu.py tests/test_transport_gpu.py tests/test_step_graphs_gpu.py tests/test_stencils_gpu.py tests/test_multirank_gpu.py -q --timeout 240 --durations=8 ) > $O/pytest.log 2>&1; echo "rc=$?"; tail -20 $O/pytest.log
