#!/bin/bash
# GPU job r5n: the plane-ring shapes compiled WITHOUT packed fp32 instructions (v_add_f32 x2 instead of v_pk_add_f32: the SQ counters of
# r5m put a v_pk_add_f32 at ~10 cycles per wave against 4 for a plain add) -- yask_amd/lib_nopk, same sources.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5n; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
export YASK_HIP_LIB_DIR=$R/yask_amd/lib_nopk
for st in cube 3plane 3axis_with_diags tti; do
  timeout 300 python3 tools/sweep_variants.py --stencil $st --size 512 --reps 5 --chunks 0 --check --steps 10 --out $O/sweep_${st}_p0.json > $O/sweep_$st.log 2>&1; grep -E "check box|WHOLE|FAILED" $O/sweep_$st.log | cut -c1-300; tail -1 $O/sweep_$st.log | cut -c1-1100
done
timeout 200 python3 tools/sweep_variants.py --stencil test_scratch_3d --part 0 --size 512 --reps 5 --chunks 0 --out $O/sweep_test_scratch_3d_p0.json > $O/sweep_ts3d.log 2>&1; tail -1 $O/sweep_ts3d.log | cut -c1-700
timeout 400 python3 tools/variant_pmc.py --stencil cube --variant box_v4_z128_y16_r1_nt_w2 --out $O/pmc_cube_r1 2>&1 | tail -3 | cut -c1-1500
