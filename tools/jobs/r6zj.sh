#!/bin/bash
# GPU job r6zj: the end of round 6 (second half): the whole GPU suite as the driver runs it, smoke, the default bench line, rocprofv3 of
# the default bench command, ssg / 3axis lines, the table of every renderable solution at 512^3, SQ counters of awp's new velocity shape.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6zj; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
( time timeout 1700 python3 -m pytest tests -m gpu -x -q --timeout 600 2>&1 | grep -v "^Solution '" ) > $O/gpu_tests.txt 2>&1
tail -n 6 $O/gpu_tests.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^Solution '" | tail -1
timeout 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; python3 -c "
import json; j=json.loads([l for l in open('$O/bench_default.json') if l.startswith('{')][0]); print('bench', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['traffic'], j['cpu_baseline']['value'])"
timeout 600 python3 tools/gpu_profile.py r6b_iso3dfd > $O/prof_iso3dfd.log 2>&1
timeout 600 python3 tools/gpu_profile.py r6b_ssg -- --workload ssg > $O/prof_ssg.log 2>&1
for t in r6b_iso3dfd r6b_ssg; do python3 -c "
import json; s=json.load(open('$R/gpurun_out/prof_$t/summary.json')); print('$t', s.get('sum_of_hot_kernel_avg_ms'), s.get('roofline_frac_at_rocprof_duration'), s.get('traffic_over_algorithmic'), {k[:70]:(v['calls'],v['avg_ms']) for k,v in s['kernels'].items()})"; rm -rf $R/gpurun_out/prof_$t/stats $R/gpurun_out/prof_$t/pmc_*/; done
timeout 300 python3 bench.py --workload 3axis --size 1024 --no-cpu-baseline > $O/bench_3axis1024.json 2>/dev/null; python3 -c "
import json; j=json.loads([l for l in open('$O/bench_3axis1024.json') if l.startswith('{')][0]); print('3axis1024', j['value'], j['ms_per_step'], j['roofline']['frac'])"
timeout 1500 python3 tools/generic_table.py --out $O --size3 512 --tag final > $O/table_final.log 2>&1; cat $O/table_final.log
timeout 600 python3 tools/variant_pmc.py --stencil awp --variant march_v4_z128_y16_nt_hr_ps_lo_w2 --part 0 --out $O/pmc_awp_p0 > $O/pmc_awp_p0.log 2>&1; tail -n 3 $O/pmc_awp_p0.log
