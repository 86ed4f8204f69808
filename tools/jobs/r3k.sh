#!/bin/bash
# GPU job r3k: after the explicit-FMA change and the _tl defaults: full GPU suite, headline bench with live traffic, rocprofv3 profile of the
# default bench command (profiles/r3k_iso3dfd), the other workloads' bench lines, decomposition cost.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3k
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -n "passed\|failed" $O/pytest.log | tail -3
echo "== headline bench, live traffic"; ( time timeout 400 python bench.py ) > $O/bench_n1.log 2>&1; echo "rc=$?"; grep '^{' $O/bench_n1.log > $O/bench_n1.json; python -c "
import json; j=json.load(open('$O/bench_n1.json')); print(j['value'], j['ms_per_step'], j['roofline'], j.get('cpu_baseline'), j['config'])" | cut -c1-1500
echo "== rocprofv3 profile of the default bench"; ( time timeout 900 python tools/gpu_profile.py r3k_iso3dfd ) > $O/gpu_profile.log 2>&1; echo "rc=$?"; tail -5 $O/gpu_profile.log | cut -c1-400; mkdir -p $O/prof_r3k_iso3dfd; cp gpurun_out/prof_r3k_iso3dfd/summary.json gpurun_out/prof_r3k_iso3dfd/kernel_stats.csv gpurun_out/prof_r3k_iso3dfd/bench_line.json gpurun_out/prof_r3k_iso3dfd/pmc_summary.json $O/prof_r3k_iso3dfd/ 2>/dev/null
echo "== other workloads"
for w in "iso3dfd --size 512" "3axis" "3axis --size 1024" "ssg" "ssg --size 1024"; do
  f="$O/bench_$(echo $w | tr ' ' '_' | tr -d '-').json"
  timeout 300 python bench.py --workload $w --no-cpu-baseline --traffic none > "$f" 2> /dev/null
  python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$w', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config'].get('kernel'))
except Exception as e: print('$w', 'ERR', e)"
done
echo "== decomp cost"; ( timeout 400 python tools/decomp_cost.py --stencil iso3dfd --quick ) > $O/decomp_iso3dfd.log 2>&1; python - <<'PY'
import json
for r in json.load(open("gpurun_out/decomp_cost_iso3dfd.json")):
    print(r["case"][:34].ljust(34), r["config"][:44].ljust(44), r["shell_or_exterior_ms"], r["rest_or_interior_ms"], r["undivided_ms"], r["overhead"], r["shell_done_at_fraction"])
PY
cp gpurun_out/decomp_cost_iso3dfd.json $O/
