#!/bin/bash
# GPU job r3o: does the pack kernel's competition for CUs explain the longer planned launch in the full schedule? (diagnostic knobs)
# (the YKH_DEBUG_SKIP_PACK / YKH_PACK_BLOCKS knobs this job used were removed from launch_halo_move() after the measurement)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3o; mkdir -p $O; cd $R
run() { tag=$1; shift; env "$@" timeout 200 python tools/overlap_probe.py --cases 2 --schedules "planned (rounds" --tag _$tag 2>&1 | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['tag'].ljust(12), r['case'][:24].ljust(24), r['ms_per_step'], r['one_rank_block_ms_per_step'], r['vs_one_rank_block'], 'ext', r['exterior_ms'], 'int', r['interior_ms'], 'pack', r['pack_ms'], 'copy', r['copy_ms'], 'unpack', r['unpack_ms'], 'wait', r['exposed_wait_ms'])
"; }
run base A=1
run skip YKH_DEBUG_SKIP_PACK=1
run b16 YKH_PACK_BLOCKS=16
run b64 YKH_PACK_BLOCKS=64
run b256 YKH_PACK_BLOCKS=256
run b1024 YKH_PACK_BLOCKS=1024
run base2 A=1
