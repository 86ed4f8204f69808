#!/bin/bash
# GPU job r6p: fused scratch kernel, three versions on ONE box, alternating: C = commit 00f598d (zero-fill, per-point tests everywhere),
# A = no zero-fill + interior-tile fast path (rolled), B = A unrolled.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6p; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R YASK_HIP_FUSE_SCRATCH=1
for rep in 1 2 3; do
  YASK_HIP_LIB_DIR=$R/yask_amd/lib_c python3 tools/generic_table.py --out $O --only swe2d wave2d --tag c$rep 2>&1 | sed "s/^/C 00f598d  rep $rep: /"
  python3 tools/generic_table.py --out $O --only swe2d wave2d --tag a$rep 2>&1 | sed "s/^/A rolled   rep $rep: /"
  YASK_HIP_LIB_DIR=$R/yask_amd/lib_b python3 tools/generic_table.py --out $O --only swe2d wave2d --tag b$rep 2>&1 | sed "s/^/B unrolled rep $rep: /"
done
