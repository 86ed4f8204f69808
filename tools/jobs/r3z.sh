#!/bin/bash
# GPU job r3z: which rank grid for 8 GPUs?  One rank of each grid (the one with the most neighbours) through the mirror transport
# under a 50 GB/s link, halves and planned + in-line pack (x-only grids keep the slab schedule)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3z; mkdir -p $O; cd $R
YASK_MIRROR_LINK_GBPS=50 timeout 150 python tools/overlap_probe.py --grid-study --tag _grids50 --schedules "halves,inline" --steps 20 2>&1 | tee $O/grids_link50.log | grep '^{' | python -c '
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r["case"][:52].ljust(52), r["schedule"][:8].ljust(8), r["ms_per_step"], r["one_rank_block_ms_per_step"], r["vs_one_rank_block"], "job Gpts/s", r["job_gpoints_per_s_at_8_ranks"], "wait", r["exposed_wait_ms"], "MB", r["halo_MB_per_step"])
'
cp gpurun_out/overlap_probe_iso3dfd_grids50.json $O/ 2>/dev/null
