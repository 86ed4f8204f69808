#!/bin/bash
# GPU job r6w: fused scratch kernel, tile shapes that let TWO workgroups share a CU (slots <= 80 KB) against the shipped 32 x 64 (one
# workgroup per CU), same box, alternating.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6w; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R YASK_HIP_FUSE_SCRATCH=1
for t in 24x40 16x56 20x48 32x32; do
  YASK_HIP_LIB_DIR=$R/yask_amd/lib_t$t timeout 300 python3 -m pytest tests/test_fused_scratch_gpu.py -m gpu -q --timeout 300 -k "match_the_reference and (wave2d or swe2d)" 2>&1 | tail -n 1
done
for rep in 1 2; do
  python3 tools/generic_table.py --out $O --only swe2d wave2d --tag s$rep 2>&1 | sed "s/^/shipped 32x64 rep $rep: /"
  for t in 24x40 16x56 20x48 32x32; do
    YASK_HIP_LIB_DIR=$R/yask_amd/lib_t$t python3 tools/generic_table.py --out $O --only swe2d wave2d --tag t${t}_$rep 2>&1 | sed "s/^/tile $t       rep $rep: /"
  done
done
