#!/bin/bash
# GPU job r6zk: the thin sub-domain parts of a stage column by column in ONE launch (ykh_column.hpp): parity, then off / on, same box.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6zk; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
( timeout 1200 python3 -m pytest tests/test_multi_tile_fixtures_gpu.py tests/test_reference_stencils_gpu.py tests/test_part_boxes_gpu.py tests/test_multirank_gpu.py tests/test_step_graphs_gpu.py -m gpu -x -q --timeout 900 2>&1 | grep -v "^Solution '" ) > $O/parity.txt 2>&1
tail -n 4 $O/parity.txt
for rep in 1 2; do
  YASK_HIP_COLUMNS=0 python3 tools/generic_table.py --out $O --only awp_abc awp_elastic_abc test_boundary_3d --size3 512 --tag off$rep 2>&1 | sed "s/^/off rep $rep: /"
  YASK_HIP_COLUMNS=1 python3 tools/generic_table.py --out $O --only awp_abc awp_elastic_abc test_boundary_3d --size3 512 --tag on$rep 2>&1 | sed "s/^/on  rep $rep: /"
done
