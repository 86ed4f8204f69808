#!/bin/bash
# GPU job r6za: fused scratch kernel, point loop prefetch vs not, same box, alternating (default tile choice).
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6za; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R YASK_HIP_FUSE_SCRATCH=1
YASK_HIP_LIB_DIR=$R/yask_amd/lib_p timeout 300 python3 -m pytest tests/test_fused_scratch_gpu.py -m gpu -q --timeout 300 -k "match_the_reference and (wave2d or swe2d)" 2>&1 | tail -n 1
for rep in 1 2 3; do
  python3 tools/generic_table.py --out $O --only swe2d wave2d --tag a$rep 2>&1 | sed "s/^/shipped  rep $rep: /"
  YASK_HIP_LIB_DIR=$R/yask_amd/lib_p python3 tools/generic_table.py --out $O --only swe2d wave2d --tag u$rep 2>&1 | sed "s/^/prefetch rep $rep: /"
done
