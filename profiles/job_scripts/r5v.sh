#!/bin/bash
# GPU job r5v: fsg's stress part, K = 4 clusters on other point-kernel shapes (8-byte lanes, two x planes per thread, 128 x 8 and 64 x 16 tiles)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5v; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
timeout 600 python3 tools/sweep_variants.py --stencil fsg --part 1 --size 512 --reps 3 --chunks 0 --check --out $O/sweep_fsg_p1.json > $O/sweep_fsg_p1.log 2>&1; grep -E "FAILED|mismatches vs naive: [1-9]" $O/sweep_fsg_p1.log | cut -c1-300; tail -n 1 $O/sweep_fsg_p1.log | cut -c1-1200; grep "'variant'" $O/sweep_fsg_p1.log | cut -c1-120
