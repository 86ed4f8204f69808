#!/bin/bash
# GPU job r3l: same-box A/B in the bench context (run_solution stepping, placement trials): _tl defaults against the plain shapes
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3l; mkdir -p $O; cd $R
run() {  # tag, bench args...
  tag=$1; shift
  timeout 200 python bench.py --no-cpu-baseline --traffic none --no-probe "$@" > $O/$tag.json 2> /dev/null
  python -c "
import json
try:
    d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); print('$tag', d['value'], d['ms_per_step'], d['step_ms']['min'], d['step_ms']['median'], d['config'].get('kernel'))
except Exception as e: print('$tag', 'ERR', e)"
}
for i in 1 2 3; do
  run iso_tl_$i
  run iso_plain_$i --opts "-hip_variant starlin_v4_z128_y32_r2_t2_nt_pd2_w2_c2"
done
for i in 1 2; do
  run iso512_tl_$i --size 512
  run iso512_plain_$i --size 512 --opts "-hip_variant starlin_v4_z128_y32_r2_t2_nt_pd2_w2_c2"
  run ax512_tl_$i --workload 3axis
  run ax512_plain_$i --workload 3axis --opts "-hip_variant starlin_v2_z64_y32_r2_u_nt_w2_c4"
done
