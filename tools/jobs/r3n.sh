#!/bin/bash
# GPU job r3n: decomposition cost tables with the final kernels (explicit FMAs, _tl twins), iso3dfd + ssg, overlap probe
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3n; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 500 python tools/decomp_cost.py --stencil iso3dfd ) > $O/decomp_iso3dfd.log 2>&1
( timeout 300 python tools/decomp_cost.py --stencil ssg ) > $O/decomp_ssg.log 2>&1
cp gpurun_out/decomp_cost_*.json $O/
python - <<'PY'
import json
for f in ("gpurun_out/decomp_cost_iso3dfd.json", "gpurun_out/decomp_cost_ssg.json"):
    for r in json.load(open(f)):
        print(r["case"][:34].ljust(34), r["config"][:50].ljust(50), r["shell_or_exterior_ms"], r["rest_or_interior_ms"], r["undivided_ms"], r["overhead"], r["shell_done_at_fraction"])
PY
( timeout 400 python tools/overlap_probe.py ) > $O/overlap.log 2>&1; tail -25 $O/overlap.log | cut -c1-300
