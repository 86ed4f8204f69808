#!/bin/bash
# GPU job r5o: tests/test_box_kernel_gpu.py (ragged multi-tile grids, tti's global-load groups, cube over 2 and 8 ranks), the
# golden parity of the solutions whose registry changed, and the no-packed build's sweeps of cube / 3axis_with_diags / tti as shipped.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5o; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
timeout 900 python3 -m pytest tests/test_box_kernel_gpu.py -m gpu -q --timeout 600 2>&1 | tail -25 > $O/box_tests.txt; tail -12 $O/box_tests.txt
timeout 600 python3 -m pytest tests/test_reference_stencils_gpu.py -m gpu -q --timeout 500 -k "cube or 3plane or 3axis_with_diags or tti or test_scratch_3d or awp_abc or awp_elastic_abc or picks_fast" 2>&1 | tail -5 > $O/parity.txt; tail -3 $O/parity.txt
for st in cube 3axis_with_diags tti; do
  timeout 300 python3 tools/sweep_variants.py --stencil $st --size 512 --reps 5 --chunks 0 --steps 10 --out $O/sweep_${st}_p0.json > $O/sweep_$st.log 2>&1; grep -E "WHOLE|FAILED" $O/sweep_$st.log | cut -c1-300
done
