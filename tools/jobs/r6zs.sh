#!/bin/bash
# GPU job r6zs: descriptor-reading twins for the generic marching shapes: decomposed runs of awp / ssg2 / iso3dfd_sponge on the planned /
# halves schedules -- parity over 2 and 8 ranks against the reference fixtures, then the compute-side cost against the undivided box.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6zs; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
( timeout 1500 python3 -m pytest tests/test_multi_tile_fixtures_gpu.py tests/test_multirank_gpu.py tests/test_decomposed_blocks_gpu.py -m gpu -x -q --timeout 900 -k "decomposed or rank" 2>&1 | grep -v "^Solution '" ) > $O/parity.txt 2>&1
tail -n 5 $O/parity.txt
sed -n '/^python3 - <<.PY./,/^PY$/p' tools/jobs/r6zr.sh | sed '1d;$d' > /tmp/slab.py
python3 /tmp/slab.py | tee $O/twin_cost.txt
