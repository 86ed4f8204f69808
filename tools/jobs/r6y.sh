#!/bin/bash
# GPU job r6y: swe2d fused, more tile shapes with two or three workgroups per CU, same box, alternating.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6y; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R YASK_HIP_FUSE_SCRATCH=1
for rep in 1 2; do
  for t in 32x32 24x24 16x32 24x32 32x24; do
    YASK_HIP_LIB_DIR=$R/yask_amd/lib_t$t python3 tools/generic_table.py --out $O --only swe2d --tag t${t}_$rep 2>&1 | sed "s/^/tile $t rep $rep: /"
  done
done
