#!/bin/bash
# GPU job r3za: halves vs planned for ssg under a 50 GB/s link, and iso3dfd under 100 GB/s (completes the tables of r3y)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3za; mkdir -p $O; cd $R
fmt='
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r["case"][:40].ljust(40), r["schedule"][:30].ljust(30), r["ms_per_step"], r["one_rank_block_ms_per_step"], r["vs_one_rank_block"], "ext", r["exterior_ms"], "int", r["interior_ms"], "pack", r["pack_ms"], "copy", r["copy_ms"], "unpack", r["unpack_ms"], "wait", r["exposed_wait_ms"])
'
YASK_MIRROR_LINK_GBPS=50 timeout 100 python tools/overlap_probe.py --stencil ssg --tag _link50 --schedules "halves,planned (rounds,inline,whole box" --steps 20 2>&1 | tee $O/ssg_link50.log | grep '^{' | python -c "$fmt"
YASK_MIRROR_LINK_GBPS=100 timeout 100 python tools/overlap_probe.py --tag _link100 --schedules "halves,planned (rounds,inline,whole box" --steps 30 2>&1 | tee $O/iso_link100.log | grep '^{' | python -c "$fmt"
cp gpurun_out/overlap_probe_ssg_link50.json gpurun_out/overlap_probe_iso3dfd_link100.json $O/ 2>/dev/null
