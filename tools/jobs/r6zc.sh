#!/bin/bash
# GPU job r6zc: experimental marching shapes (-DYKH_MARCH_EXP: two workgroups per CU, planes two ahead, late refill of the once operands)
# on awp / awp_elastic / ssg2, every shape checked against the point kernel.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6zc; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R YASK_HIP_LIB_DIR=$R/yask_amd/lib_x
for s in awp awp_elastic ssg2; do
  for p in 0 1; do
    timeout 400 python3 tools/sweep_variants.py --stencil $s --size 512 --part $p --chunks 0 --reps 5 --check --out $O/sweep_${s}_p$p.json > $O/sweep_${s}_p$p.log 2>&1
    grep -v "^BEST" $O/sweep_${s}_p$p.log | tail -n 30
  done
done
