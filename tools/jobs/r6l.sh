#!/bin/bash
# GPU job r6l: per-box shape choice for parts that run over a box list (shells, 2-D ring strips): parity, then what it buys.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6l; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
( time timeout 1200 python3 -m pytest tests/test_part_boxes_gpu.py tests/test_multi_tile_fixtures_gpu.py tests/test_fused_scratch_gpu.py tests/test_reference_stencils_gpu.py tests/test_clusters_gpu.py -m gpu -q --timeout 600 2>&1 | grep -v "^Solution '" ) > $O/tests.txt 2>&1
tail -n 6 $O/tests.txt
python3 tools/generic_table.py --out $O --only fsg_abc fsg2_abc fsg_merged_abc test_boundary_3d awp_abc awp_elastic_abc --size3 512 --tag shells512 > $O/shells.log 2>&1; cat $O/shells.log
YASK_HIP_FUSE_SCRATCH=0 python3 tools/generic_table.py --out $O --only swe2d wave2d test_boundary_2d --tag rings_unfused > $O/rings.log 2>&1; cat $O/rings.log
