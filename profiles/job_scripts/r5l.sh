#!/bin/bash
# GPU job r5l: plane-ring kernel, second shapes (two rows per thread as one wide vector, planes requested two iterations ahead, one
# wave per SIMD for tti): parity of every variant against the reference's outputs, every shape timed at 512^3 and checked against
# the point kernel; wave-front tiling across ranks (-Mbt 2 / 3) against the plain schedules for ssg under an emulated 50 and 25 GB/s link.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5l; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
timeout 600 python3 -m pytest tests/test_reference_stencils_gpu.py -m gpu -q --timeout 500 -k "cube or 3plane or 3axis_with_diags or tti or test_scratch_3d or awp_abc or awp_elastic_abc or picks_fast" 2>&1 | tail -25 > $O/parity.txt; tail -8 $O/parity.txt
for st in cube 3plane 3axis_with_diags tti; do
  timeout 300 python3 tools/sweep_variants.py --stencil $st --size 512 --reps 5 --chunks 0 --check --steps 10 --out $O/sweep_${st}_p0.json > $O/sweep_$st.log 2>&1; grep -E "check box|WHOLE|FAILED" $O/sweep_$st.log | cut -c1-300; tail -1 $O/sweep_$st.log | cut -c1-1100
done
timeout 200 python3 tools/sweep_variants.py --stencil test_scratch_3d --part 0 --size 512 --reps 5 --chunks 0 --out $O/sweep_test_scratch_3d_p0.json > $O/sweep_ts3d.log 2>&1; tail -1 $O/sweep_ts3d.log | cut -c1-700
for g in 50 25; do
  YASK_MIRROR_LINK_GBPS=$g timeout 400 python3 tools/overlap_probe.py --stencil ssg --mbt --tag _mbt_link$g > $O/mbt_ssg_link$g.log 2>&1; cp gpurun_out/overlap_probe_ssg_mbt_link$g.json $O/ 2>/dev/null
  grep -E '"schedule"' $O/mbt_ssg_link$g.log | python3 -c "
import sys,json
for l in sys.stdin:
    try: j=json.loads(l)
    except Exception: continue
    print('$g GB/s |', j.get('case','')[:28], '|', j.get('schedule','')[:40], '|', j.get('ms_per_step'), j.get('vs_one_rank_block'), 'wait', j.get('exposed_wait_ms'), j.get('error','')[:100])"
done
