#!/bin/bash
# GPU job r5u: tti with the 8-byte-lane plane-ring shape (two waves per SIMD, no spill): parity + every shape timed at 512^3.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5u; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
timeout 400 python3 -m pytest tests/test_reference_stencils_gpu.py tests/test_box_kernel_gpu.py -m gpu -q --timeout 300 -k "tti" 2>&1 | tail -4 > $O/parity.txt; tail -n 2 $O/parity.txt
timeout 300 python3 tools/sweep_variants.py --stencil tti --size 512 --reps 5 --chunks 0 --check --steps 10 --out $O/sweep_tti_p0.json > $O/sweep_tti.log 2>&1; grep -E "WHOLE|FAILED|mismatches vs naive: [1-9]" $O/sweep_tti.log | cut -c1-300; tail -n 1 $O/sweep_tti.log | cut -c1-700
