#!/bin/bash
# GPU job r6k: evidence for the end of round 6 -- the whole GPU suite as the driver runs it, smoke, then rocprofv3 (tools/gpu_profile.py:
# --kernel-trace --stats of the default bench command + separate --pmc passes) for the headline, 3axis fp64 1024^3 and ssg 512^3.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6k; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
( time timeout 1700 python3 -m pytest tests -m gpu -x -q --timeout 600 2>&1 | grep -v "^Solution '" ) > $O/gpu_tests.txt 2>&1
tail -n 6 $O/gpu_tests.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^Solution '" | tail -1
timeout 600 python3 tools/gpu_profile.py r6_iso3dfd > $O/prof_iso3dfd.log 2>&1; tail -30 $O/prof_iso3dfd.log | head -5
timeout 600 python3 tools/gpu_profile.py r6_3axis1024 -- --workload 3axis --size 1024 > $O/prof_3axis1024.log 2>&1
timeout 600 python3 tools/gpu_profile.py r6_ssg -- --workload ssg > $O/prof_ssg.log 2>&1
for t in r6_iso3dfd r6_3axis1024 r6_ssg; do python3 -c "
import json; s=json.load(open('$R/gpurun_out/prof_$t/summary.json')); print('$t', s.get('sum_of_hot_kernel_avg_ms'), s.get('roofline_frac_at_rocprof_duration'), s.get('traffic_over_algorithmic'), {k[:70]:(v['calls'],v['avg_ms']) for k,v in s['kernels'].items()})"; rm -rf $R/gpurun_out/prof_$t/stats $R/gpurun_out/prof_$t/pmc_*/; done
timeout 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; python3 -c "
import json; j=json.loads([l for l in open('$O/bench_default.json') if l.startswith('{')][0]); print('bench', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['traffic'], j['cpu_baseline']['value'])"
