#!/bin/bash
# GPU job r6q: fused scratch kernel on ONE box, alternating: C = commit 00f598d (zero-fill per tile), D = C without the zero-fill.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6q; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R YASK_HIP_FUSE_SCRATCH=1
YASK_HIP_LIB_DIR=$R/yask_amd/lib_d timeout 600 python3 -m pytest tests/test_fused_scratch_gpu.py -m gpu -q --timeout 300 -k "wave2d or swe2d" 2>&1 | tail -n 2
for rep in 1 2 3; do
  YASK_HIP_LIB_DIR=$R/yask_amd/lib_c python3 tools/generic_table.py --out $O --only swe2d wave2d --tag c$rep 2>&1 | sed "s/^/C 00f598d      rep $rep: /"
  YASK_HIP_LIB_DIR=$R/yask_amd/lib_d python3 tools/generic_table.py --out $O --only swe2d wave2d --tag d$rep 2>&1 | sed "s/^/D no zero-fill rep $rep: /"
done
