#!/bin/bash
# GPU job r6zo: test_4d with the 3-D families at the table's size (16 x 256^3), the one-tile fixtures of every solution on the final libraries.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6zo; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
python3 tools/generic_table.py --out $O --only test_4d --size3 512 --tag t4d 2>&1
( timeout 900 python3 -m pytest tests/test_reference_stencils_gpu.py tests/test_python_api_gpu.py -m gpu -x -q --timeout 600 2>&1 | grep -v "^Solution '" ) > $O/parity.txt 2>&1
tail -n 3 $O/parity.txt
