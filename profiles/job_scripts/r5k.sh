#!/bin/bash
# GPU job r5k: first run of the plane-ring kernel (csrc/ykh_box.hpp): parity of every variant of the affected solutions against the
# reference's outputs, then every shape of cube / 3plane / 3axis_with_diags / tti / test_scratch_3d timed at 512^3 and checked against
# the point kernel at full size.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5k; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
timeout 600 python3 -m pytest tests/test_reference_stencils_gpu.py -m gpu -q --timeout 500 -k "cube or 3plane or 3axis_with_diags or tti or test_scratch_3d or awp_abc or awp_elastic_abc or picks_fast" 2>&1 | tail -25 > $O/parity.txt; tail -8 $O/parity.txt
for st in cube 3plane 3axis_with_diags tti; do
  timeout 300 python3 tools/sweep_variants.py --stencil $st --size 512 --reps 5 --chunks 0 --check --steps 10 --out $O/sweep_${st}_p0.json > $O/sweep_$st.log 2>&1; grep -E "check|WHOLE|FAILED" $O/sweep_$st.log | cut -c1-300; tail -1 $O/sweep_$st.log | cut -c1-900
done
timeout 200 python3 tools/sweep_variants.py --stencil test_scratch_3d --part 0 --size 512 --reps 5 --chunks 0 --out $O/sweep_test_scratch_3d_p0.json > $O/sweep_ts3d.log 2>&1; tail -1 $O/sweep_ts3d.log | cut -c1-700
