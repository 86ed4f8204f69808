#!/bin/bash
# GPU job r6zf: thin sub-domain parts of a stage side by side on their own streams (awp_abc's free-surface planes): parity against
# the reference fixtures, then the step time with the streams off / on, same box, alternating.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6zf; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
( YASK_HIP_THIN_STREAMS=1 timeout 900 python3 -m pytest tests/test_multi_tile_fixtures_gpu.py tests/test_reference_stencils_gpu.py tests/test_part_boxes_gpu.py -m gpu -x -q --timeout 900 -k "awp or boundary or abc" 2>&1 | grep -v "^Solution '" ) > $O/parity_on.txt 2>&1
tail -n 3 $O/parity_on.txt
for rep in 1 2; do
  YASK_HIP_THIN_STREAMS=0 python3 tools/generic_table.py --out $O --only awp_abc awp_elastic_abc --size3 512 --tag off$rep 2>&1 | sed "s/^/off rep $rep: /"
  YASK_HIP_THIN_STREAMS=1 python3 tools/generic_table.py --out $O --only awp_abc awp_elastic_abc --size3 512 --tag on$rep 2>&1 | sed "s/^/on  rep $rep: /"
done
