#!/bin/bash
# GPU job r3r: final evidence of the round for the decomposed path: full gpu suite, overlap probe (iso3dfd + ssg), decomposition cost
# (iso3dfd + ssg), rocprofv3 kernel trace (time line) of a decomposed rank on the mirror transport.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3r; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1000 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; grep -n "passed\|failed" $O/pytest.log | tail -3
( timeout 300 python tools/overlap_probe.py --stencil iso3dfd ) > $O/overlap_iso3dfd.log 2>&1; cp gpurun_out/overlap_probe_iso3dfd.json $O/
( timeout 300 python tools/overlap_probe.py --stencil ssg ) > $O/overlap_ssg.log 2>&1; cp gpurun_out/overlap_probe_ssg.json $O/
( timeout 400 python tools/decomp_cost.py --stencil iso3dfd ) > $O/decomp_iso3dfd.log 2>&1
( timeout 300 python tools/decomp_cost.py --stencil ssg ) > $O/decomp_ssg.log 2>&1
cp gpurun_out/decomp_cost_*.json $O/
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_decomp -- python $R/tools/overlap_probe.py --stencil iso3dfd --steps 30 --cases 1 --schedules "planned (rounds" ) > $O/prof_decomp.log 2>&1; echo "rocprof rc=$?"
cp $(find $O/prof_decomp -name "*kernel_stats.csv" | head -1) $O/decomp_kernel_stats.csv 2>/dev/null
python - <<'PY'
import json
for f in ("gpurun_out/r3r/overlap_probe_iso3dfd.json","gpurun_out/r3r/overlap_probe_ssg.json"):
    for r in json.load(open(f)):
        print(r['case'][:26].ljust(26), r['schedule'][:40].ljust(40), r['ms_per_step'], r['vs_one_rank_block'], r['pack_ms'], r['copy_ms'], r['unpack_ms'], r['exposed_wait_ms'])
for f in ("gpurun_out/decomp_cost_iso3dfd.json", "gpurun_out/decomp_cost_ssg.json"):
    for r in json.load(open(f)):
        print(r["case"][:34].ljust(34), r["config"][:50].ljust(50), r["shell_or_exterior_ms"], r["rest_or_interior_ms"], r["undivided_ms"], r["overhead"], r["shell_done_at_fraction"])
PY
