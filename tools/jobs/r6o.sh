#!/bin/bash
# GPU job r6o: fused scratch kernel A/B on ONE box, alternating: the shipped loop (one iteration at a time) vs the unrolled one.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6o; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R YASK_HIP_FUSE_SCRATCH=1
timeout 600 python3 -m pytest tests/test_fused_scratch_gpu.py -m gpu -q --timeout 300 2>&1 | tail -n 2
YASK_HIP_LIB_DIR=$R/yask_amd/lib_b timeout 600 python3 -m pytest tests/test_fused_scratch_gpu.py -m gpu -q --timeout 300 -k "wave2d or swe2d" 2>&1 | tail -n 2
for rep in 1 2 3; do
  python3 tools/generic_table.py --out $O --only swe2d wave2d --tag a$rep 2>&1 | sed "s/^/A rolled   rep $rep: /"
  YASK_HIP_LIB_DIR=$R/yask_amd/lib_b python3 tools/generic_table.py --out $O --only swe2d wave2d --tag b$rep 2>&1 | sed "s/^/B unrolled rep $rep: /"
done
