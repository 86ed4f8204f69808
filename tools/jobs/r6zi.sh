#!/bin/bash
# GPU job r6zi: compile-time dims of partial-dim tables in the point kernels and the plane-ring kernel (test_partial_3d): parity, table, shapes.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6zi; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
( time timeout 1500 python3 -m pytest tests/test_multi_tile_fixtures_gpu.py tests/test_reference_stencils_gpu.py tests/test_compile_time_variants_gpu.py tests/test_clusters_gpu.py tests/test_part_boxes_gpu.py tests/test_box_kernel_gpu.py -m gpu -x -q --timeout 900 2>&1 | grep -v "^Solution '" ) > $O/parity.txt 2>&1
tail -n 6 $O/parity.txt
python3 tools/generic_table.py --out $O --only test_partial_3d awp_abc awp_elastic_abc test_boundary_3d fsg_abc --size3 512 --tag dims2 > $O/dims2.log 2>&1; cat $O/dims2.log
timeout 400 python3 tools/sweep_variants.py --stencil test_partial_3d --size 512 --part 0 --chunks 0 --reps 5 --out $O/sweep_tp3d.json > $O/sweep_tp3d.log 2>&1
echo "== test_partial_3d"; grep "^{'variant'" $O/sweep_tp3d.log | sed "s/'xchunk': 0, //; s/, 'gpoints.*//" | sort -t: -k3 -n | head -n 12
