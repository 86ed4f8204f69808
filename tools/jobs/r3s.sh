#!/bin/bash
# GPU job r3s: planned twin at 242 VGPRs (room for a waiter wave beside it): decomposition cost, overlap probe incl. the one-launch + signal form
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3s; mkdir -p $O; cd $R
timeout 300 python tools/decomp_cost.py --stencil iso3dfd --quick 2>&1 | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['case'][:30].ljust(30), r['config'][:60].ljust(60), r['shell_or_exterior_ms'], r['rest_or_interior_ms'], r['undivided_ms'], r['overhead'], r['shell_done_at_fraction'])
"
timeout 300 python tools/overlap_probe.py --schedules planned 2>&1 | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['case'][:24].ljust(24), r['schedule'][:34].ljust(34), r['ms_per_step'], r['one_rank_block_ms_per_step'], r['vs_one_rank_block'], 'ext', r['exterior_ms'], 'int', r['interior_ms'], 'pack', r['pack_ms'], 'copy', r['copy_ms'], 'unpack', r['unpack_ms'], 'wait', r['exposed_wait_ms'])
"
cp gpurun_out/overlap_probe_iso3dfd.json gpurun_out/decomp_cost_iso3dfd.json $O/
timeout 600 python -m pytest tests/test_transport_gpu.py tests/test_decomposed_blocks_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; grep -n "passed\|failed" $O/pytest.log | tail -3
