#!/bin/bash
# GPU job r4l: every compiled iso3dfd shape (YKH_PROFILING build, yask_amd/lib_prof) at 512^3 with the automatic x-chunking and
# chunks of 128 / 256 / 512 planes -- is there a shape for THIS size that beats the 128x32 default?
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4l; mkdir -p $O; cd $R
YASK_HIP_LIB_DIR=$R/yask_amd/lib_prof timeout 400 python tools/sweep_variants.py --stencil iso3dfd --size 512 --reps 30 --chunks 0 128 256 512 --out $O/sweep_iso3dfd_512.json > $O/sweep.log 2>&1
grep -v "abl" $O/sweep.log | grep "^{" | sort -t: -k4 -n | python -c "
import sys, ast
rows = [ast.literal_eval(l) for l in sys.stdin if l.startswith('{')]
rows.sort(key=lambda r: r['ms'])
for r in rows[:24]: print(r)
"
# the step-timer question: wall ms per step at 512^3 with and without the per-step events
for o in "" "-no-hip_step_timers"; do timeout 100 python bench.py --size 512 --steps 200 --no-cpu-baseline --no-probe --traffic none "--opts=$o" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('512^3 [$o]', j['ms_per_step'], 'ms/step', j['value'])"; done
