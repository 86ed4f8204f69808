#!/bin/bash
# GPU job r4n: does the placement lottery follow physical placement, and can the virtual-memory API (hipMemCreate / hipMemMap) choose it?
# tools/microbench/vmm_placement.hip (written at the end of round 3)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4n; mkdir -p $O; cd $R/tools/microbench
hipcc -O3 --offload-arch=gfx950 vmm_placement.hip -o /tmp/vmm_placement 2>&1 | tail -3
timeout 200 /tmp/vmm_placement 1024 64 6 8 2>&1 | tee $O/vmm_placement_64MiB.txt | tail -45
timeout 100 /tmp/vmm_placement 1024 2 2 8 2>&1 | tee $O/vmm_placement_2MiB.txt | tail -22
