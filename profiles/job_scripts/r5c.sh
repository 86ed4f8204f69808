#!/bin/bash
# GPU job r5c: a roofline for every renderable solution (tools/generic_table.py) + the reference's own Python
# API scripts run unchanged (tests/test_reference_api_programs_gpu.py) + the changed C API entry point.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5c; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python3 -m pytest tests/test_reference_api_programs_gpu.py tests/test_python_api_gpu.py -m gpu -q --timeout 300 2>&1 | tail -25 > $O/ref_programs.txt; tail -12 $O/ref_programs.txt
timeout 900 python3 tools/generic_table.py --out $O --size3 256 > $O/table_256.log 2>&1; tail -60 $O/table_256.log
timeout 400 python3 tools/generic_table.py --out $O --size3 512 --tag table512 --only iso3dfd 3axis ssg ssg2 fsg fsg2 awp awp_elastic tti iso3dfd_sponge ssg_merged fsg_merged cube 3plane test_3d >> $O/table_512.log 2>&1; tail -20 $O/table_512.log
