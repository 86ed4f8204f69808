#!/bin/bash
# GPU job r6ze: late-refill (_lo) shapes in the generic registry: parity of every registered shape against the reference fixtures,
# then the table of the solutions they touch.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6ze; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
( time timeout 1500 python3 -m pytest tests/test_multi_tile_fixtures_gpu.py tests/test_reference_stencils_gpu.py tests/test_compile_time_variants_gpu.py tests/test_clusters_gpu.py tests/test_part_boxes_gpu.py -m gpu -x -q --timeout 900 2>&1 | grep -v "^Solution '" ) > $O/parity.txt 2>&1
tail -n 6 $O/parity.txt
python3 tools/generic_table.py --out $O --only awp awp_abc awp_elastic awp_elastic_abc ssg2 ssg_merged iso3dfd_sponge fsg --size3 512 --tag lo > $O/lo.log 2>&1; cat $O/lo.log
