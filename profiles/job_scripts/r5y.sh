#!/bin/bash
# GPU job r5y: find_part_boxes after its host logic moved to ykh_boxes.hpp: the geometry tests + the golden parity of the conditional solutions.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5y; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
timeout 60 python3 -m pytest tests/test_part_boxes_gpu.py tests/test_reference_stencils_gpu.py -m gpu -q --timeout 50 -k "boxes or boundary_3d or fsg_abc or awp_abc or sub_domain" 2>&1 | tail -n 4 > $O/tests.txt; cat $O/tests.txt
