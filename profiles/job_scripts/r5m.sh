#!/bin/bash
# GPU job r5m: where do the plane-ring kernel's cycles go?  SQ counters of cube's one-row and two-row shapes and of the point kernel.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5m; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
for v in box_v4_z128_y16_r1_nt_w2 box_v4_z128_y16_r2_nt_p2_w2 vecpt_v4_z256_y4_x1; do
  timeout 400 python3 tools/variant_pmc.py --stencil cube --variant $v --out $O/pmc_cube_$v 2>&1 | tail -6 | cut -c1-1500
done
