#!/bin/bash
# GPU job r6n: the whole GPU suite on the tree after the per-box shape choice, the 8-byte-lane vector point kernel and the cheaper
# addresses of partial-dim reads; test_partial_3d / iso3dfd_sponge / awp_abc tables.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6n; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
( time timeout 1700 python3 -m pytest tests -m gpu -x -q --timeout 600 2>&1 | grep -v "^Solution '" ) > $O/gpu_tests.txt 2>&1
tail -n 6 $O/gpu_tests.txt
python3 tools/generic_table.py --out $O --only test_partial_3d iso3dfd_sponge awp_abc test_scratch_3d cube tti --size3 512 --tag after > $O/after.log 2>&1; cat $O/after.log
