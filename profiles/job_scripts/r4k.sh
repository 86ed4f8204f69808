#!/bin/bash
# GPU job r4k: transport / multirank tests after hipMemcpyDefault + waiter early exit; temporal blocking where it pays (the heat3d
# reading of BASELINE config 3): plain sweeps against two steps per pass, same box, same process settings
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4k; mkdir -p $O; cd $R
timeout 400 python -m pytest tests/test_transport_gpu.py tests/test_multirank_gpu.py tests/test_fused_gpu.py -m gpu -q 2>&1 > $O/tests.log; grep -E "passed|failed|error" $O/tests.log | tail -3
for n in 512 1024; do for o in "" "-hip_fuse_steps 2"; do
  timeout 120 python bench.py --workload heat3d --size $n --steps 50 --no-cpu-baseline --no-probe --traffic none "--opts=$o" 2>/dev/null > $O/heat3d_${n}_$(echo $o | tr -d ' -').json
  python -c "
import json; j=json.load(open('$O/heat3d_${n}_$(echo $o | tr -d ' -').json')); print('heat3d $n [$o]', j['value'], 'Gpoints/s', j['ms_per_step'], 'ms/step, frac', j['roofline']['frac'], 'fused passes', j['config']['fused_two_step_passes_in_timed_region'], j['config']['kernel'])"
done; done
