#!/bin/bash
# GPU job r6zh: partial-dim operands by their compile-time dims (group_dims: hoisted / one value per row) in the marching and
# linear-star kernels: parity of every registered shape, then the table of the solutions that have such operands.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6zh; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
( time timeout 1500 python3 -m pytest tests/test_multi_tile_fixtures_gpu.py tests/test_reference_stencils_gpu.py tests/test_compile_time_variants_gpu.py tests/test_clusters_gpu.py tests/test_part_boxes_gpu.py tests/test_iso3dfd_gpu.py tests/test_stencils_gpu.py -m gpu -x -q --timeout 900 2>&1 | grep -v "^Solution '" ) > $O/parity.txt 2>&1
tail -n 6 $O/parity.txt
python3 tools/generic_table.py --out $O --only awp awp_abc awp_elastic awp_elastic_abc iso3dfd_sponge test_partial_3d ssg2 --size3 512 --tag dims > $O/dims.log 2>&1; cat $O/dims.log
for p in 0 1; do
  timeout 400 python3 tools/sweep_variants.py --stencil awp --size 512 --part $p --chunks 0 --reps 5 --out $O/sweep_awp_p$p.json > $O/sweep_awp_p$p.log 2>&1
  echo "== awp part $p"; grep "^{'variant'" $O/sweep_awp_p$p.log | sed "s/'xchunk': 0, //; s/, 'gpoints.*//" | sort -t: -k3 -n | head -n 8
done
timeout 400 python3 tools/sweep_variants.py --stencil iso3dfd_sponge --size 512 --part 0 --chunks 0 --reps 5 --out $O/sweep_sponge.json > $O/sweep_sponge.log 2>&1
echo "== sponge"; grep "^{'variant'" $O/sweep_sponge.log | sed "s/'xchunk': 0, //; s/, 'gpoints.*//" | sort -t: -k3 -n | head -n 8
