#!/bin/bash
# GPU job r3f: evidence for DESIGN / profiles: validate table, ssg fused-traffic microbenchmark, decomposition cost (final defaults +
# wave-front multi), headline bench with live traffic, rocprofv3 profile of the default bench command, power probe.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3f
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== validate table"; ( time timeout 400 python tools/validate_table.py ) > $O/validate.log 2>&1; echo "rc=$?"; grep '^{' $O/validate.log | cut -c1-420; cp gpurun_out/validate_table.json $O/ 2>/dev/null
echo "== ssg fused traffic"; ( time timeout 200 tools/microbench/ssg_fused_traffic 512 ) > $O/ssg_fused_traffic.txt 2>&1; echo "rc=$?"; cat $O/ssg_fused_traffic.txt | head -8
echo "== decomp cost"; ( timeout 400 python tools/decomp_cost.py --stencil iso3dfd ) > $O/decomp_iso3dfd.log 2>&1; ( timeout 300 python tools/decomp_cost.py --stencil ssg ) > $O/decomp_ssg.log 2>&1; cp gpurun_out/decomp_cost_*.json $O/; python - <<'PY'
import json
for f in ("gpurun_out/decomp_cost_iso3dfd.json", "gpurun_out/decomp_cost_ssg.json"):
    for r in json.load(open(f)):
        print(r["case"][:34].ljust(34), r["config"][:44].ljust(44), r["shell_or_exterior_ms"], r["rest_or_interior_ms"], r["undivided_ms"], r["overhead"], r["shell_done_at_fraction"])
PY
echo "== headline bench, live traffic"; ( time timeout 400 python bench.py ) > $O/bench_n1.log 2>&1; echo "rc=$?"; grep '^{' $O/bench_n1.log > $O/bench_n1.json; python -c "
import json; j=json.load(open('$O/bench_n1.json')); print(j['value'], j['ms_per_step'], j['roofline'], j.get('cpu_baseline'))" | cut -c1-1200
echo "== rocprofv3 profile of the default bench"; ( time timeout 900 python tools/gpu_profile.py r3f_iso3dfd ) > $O/gpu_profile.log 2>&1; echo "rc=$?"; tail -5 $O/gpu_profile.log | cut -c1-400; mkdir -p $O/prof_r3f_iso3dfd; cp gpurun_out/prof_r3f_iso3dfd/summary.json gpurun_out/prof_r3f_iso3dfd/kernel_stats.csv gpurun_out/prof_r3f_iso3dfd/bench_line.json gpurun_out/prof_r3f_iso3dfd/pmc_summary.json $O/prof_r3f_iso3dfd/ 2>/dev/null
echo "== power probe"; ( time timeout 900 bash tools/power_probe.sh $O/power ) > $O/power.log 2>&1; echo "rc=$?"; cat $O/power.log | tail -12
