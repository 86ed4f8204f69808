#!/bin/bash
# GPU job r6zu: descriptor-reading twins for the plane-ring shapes too: cost against the undivided box, then the whole GPU suite
# (cube / tti over 2 and 8 ranks against the reference fixtures among it).
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6zu; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
sed -n '/^python3 - <<.PY./,/^PY$/p' tools/jobs/r6zr.sh | sed '1d;$d' > /tmp/slab.py
python3 /tmp/slab.py | tee $O/twin_cost.txt
( time timeout 1700 python3 -m pytest tests -m gpu -x -q --timeout 600 2>&1 | grep -v "^Solution '" ) > $O/gpu_tests.txt 2>&1
grep -n "passed\|failed\|Error" $O/gpu_tests.txt | tail -5
