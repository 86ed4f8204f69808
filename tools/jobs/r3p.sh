#!/bin/bash
# GPU job r3p: planned launches as two launches + event (no resident waiter) against one launch + signal: overlap probe, decomp cost, tests
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3p; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python tools/overlap_probe.py --schedules planned 2>&1 | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['case'][:24].ljust(24), r['schedule'][:34].ljust(34), r['ms_per_step'], r['one_rank_block_ms_per_step'], r['vs_one_rank_block'], 'ext', r['exterior_ms'], 'int', r['interior_ms'], 'pack', r['pack_ms'], 'copy', r['copy_ms'], 'unpack', r['unpack_ms'], 'wait', r['exposed_wait_ms'])
"
cp gpurun_out/overlap_probe_iso3dfd.json $O/
timeout 300 python tools/decomp_cost.py --stencil iso3dfd --quick 2>&1 | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['case'][:30].ljust(30), r['config'][:60].ljust(60), r['shell_or_exterior_ms'], r['rest_or_interior_ms'], r['undivided_ms'], r['overhead'], r['shell_done_at_fraction'])
"
timeout 900 python -m pytest tests/test_transport_gpu.py tests/test_decomposed_blocks_gpu.py tests/test_multirank_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; grep -n "passed\|failed\|Error" $O/pytest.log | tail -5
