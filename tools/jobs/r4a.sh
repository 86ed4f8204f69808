#!/bin/bash
# GPU job r4a (prepared at the end of round 3, not yet run): does the placement lottery follow physical placement, and can the
# virtual-memory API (hipMemCreate / hipMemMap) choose it?  tools/microbench/vmm_placement.hip
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4a; mkdir -p $O; cd $R/tools/microbench
[ -x vmm_placement ] || hipcc -O3 --offload-arch=gfx950 vmm_placement.hip -o vmm_placement
timeout 240 ./vmm_placement 1024 64 6 8 2>&1 | tee $O/vmm_placement_64MiB.txt | tail -40
timeout 120 ./vmm_placement 1024 2 2 8 2>&1 | tee $O/vmm_placement_2MiB.txt | tail -20
