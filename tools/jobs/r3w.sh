#!/bin/bash
# GPU job r3w: new mirror without its allocation at link 0: are the later schedules of one process still slow?
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3w; mkdir -p $O; cd $R
timeout 300 python tools/overlap_probe.py --cases 1 --tag _w0 2>&1 | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['schedule'][:34].ljust(34), r['ms_per_step'], r['vs_one_rank_block'], 'ext', r['exterior_ms'], 'int', r['interior_ms'], 'pack', r['pack_ms'], 'copy', r['copy_ms'], 'unpack', r['unpack_ms'], 'wait', r['exposed_wait_ms'])
"
echo "== serial first, then planned (order reversed)"
timeout 300 python tools/overlap_probe.py --cases 1 --schedules "whole box" --tag _w1 2>&1 | grep '^{' | cut -c1-230
