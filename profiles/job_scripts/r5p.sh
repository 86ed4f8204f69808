#!/bin/bash
# GPU job r5p: fsg's two parts as clusters of equations (csrc/ykh_subpart.hpp): every registered shape against the reference's
# outputs (golden grids), then all shapes of both parts timed at 512^3 and checked against the point kernel.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5p; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
timeout 900 python3 -m pytest tests/test_reference_stencils_gpu.py -m gpu -q --timeout 800 -k "fsg" 2>&1 | tail -15 > $O/parity.txt; tail -6 $O/parity.txt
for p in 0 1; do
  timeout 600 python3 tools/sweep_variants.py --stencil fsg --part $p --size 512 --reps 3 --chunks 0 --check $([ $p = 1 ] && echo --steps 6) --out $O/sweep_fsg_p$p.json > $O/sweep_fsg_p$p.log 2>&1; grep -E "check c|WHOLE|FAILED" $O/sweep_fsg_p$p.log | cut -c1-300; tail -1 $O/sweep_fsg_p$p.log | cut -c1-1300
done
