#!/bin/bash
# GPU job r5j: rocprofv3 evidence for round 5 (tools/gpu_profile.py: --kernel-trace --stats of the default bench command + separate --pmc
# passes) for the headline, for 3axis fp64 1024^3 on its new default shape and for ssg 512^3; `python3 bench.py --gpus 2` started plainly
# (it launches its own ranks; gloo: both on this one device); the test fixed after r5i.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5j; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
timeout 200 python3 -m pytest tests/test_reference_stencils_gpu.py -m gpu -q --timeout 300 2>&1 | tail -4
YASK_DIST_BACKEND=gloo timeout 400 python3 bench.py --gpus 2 --size 512 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_selflaunch_n2_gloo.json 2> $O/bench_selflaunch_n2_gloo.err; echo "self-launch rc=$?"; python3 -c "
import json,sys
j=json.loads(open('$O/bench_selflaunch_n2_gloo.json').read().strip().splitlines()[-1]); print('n_gpus',j['n_gpus'],'value',j['value'],'schedule',j['config']['schedule'],j['config']['schedule_trials_ms_per_step'],'transport',j['config']['halo_transport'],j['config']['self_check']['transports'] if j['config'].get('self_check') else None)" 2>&1 | cut -c1-900
timeout 600 python3 tools/gpu_profile.py r5_iso3dfd > $O/prof_iso3dfd.log 2>&1; tail -30 $O/prof_iso3dfd.log | head -5
timeout 600 python3 tools/gpu_profile.py r5_3axis1024 -- --workload 3axis --size 1024 > $O/prof_3axis1024.log 2>&1
timeout 600 python3 tools/gpu_profile.py r5_ssg -- --workload ssg > $O/prof_ssg.log 2>&1
for t in r5_iso3dfd r5_3axis1024 r5_ssg; do python3 -c "
import json; s=json.load(open('$R/gpurun_out/prof_$t/summary.json')); print('$t', s.get('sum_of_hot_kernel_avg_ms'), s.get('roofline_frac_at_rocprof_duration'), s.get('traffic_over_algorithmic'), {k[:70]:(v['calls'],v['avg_ms']) for k,v in s['kernels'].items()})"; rm -rf $R/gpurun_out/prof_$t/stats $R/gpurun_out/prof_$t/pmc_*/; done
