#!/bin/bash
# GPU job r6g: fused scratch groups with 32-bit offsets (16 x 64 tile) and, from an experiment library, the 32 x 64 tile; forced on.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6g; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
timeout 600 python3 -m pytest tests/test_fused_scratch_gpu.py -m gpu -q --timeout 300 -k "match_the_reference" 2>&1 | tail -n 3
YASK_HIP_FUSE_SCRATCH=1 python3 tools/generic_table.py --out $O --only swe2d wave2d wave2d_f64 test_scratch_2d --tag fused_16x64 > $O/a.log 2>&1; cat $O/a.log
YASK_HIP_LIB_DIR=$R/yask_amd/lib_t32 YASK_HIP_FUSE_SCRATCH=1 python3 tools/generic_table.py --out $O --only swe2d wave2d --tag fused_32x64 > $O/b.log 2>&1; cat $O/b.log
