#!/bin/bash
# GPU job r5h: lock-step A/B on the remaining hot kernels (3axis 512^3 default shape, ssg's two stages at 512^3), re-check of the
# shipped 3axis shape after the helper refactoring, the full GPU suite, the default bench line (third box with hot placement trials),
# the final generic tables.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5h; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
P=$R/yask_amd/lib_prof
YASK_HIP_LIB_DIR=$P timeout 200 python3 tools/lockstep_probe.py --stencil 3axis --only 64 --passes 3 > $O/ls_3axis_1024.log 2>&1; tail -2 $O/ls_3axis_1024.log | cut -c1-500
YASK_HIP_LIB_DIR=$P timeout 200 python3 tools/lockstep_probe.py --stencil 3axis512 --size 512 --only 16 32 64 --passes 4 > $O/ls_3axis_512.log 2>&1; tail -2 $O/ls_3axis_512.log | cut -c1-700
S1=march_v4_z128_y16_nt_hr_ps_fd_t2
YASK_HIP_LIB_DIR=$P timeout 300 python3 tools/lockstep_probe.py --stencil ssg --part 0 --size 512 --passes 4 --shapes ${S1}_w2 ${S1}_ls16_w2 ${S1}_ls64_w2 > $O/ls_ssg_p0_512.log 2>&1; tail -2 $O/ls_ssg_p0_512.log | cut -c1-700
S2=march_v4_z128_y16_nt_hr_fd
YASK_HIP_LIB_DIR=$P timeout 300 python3 tools/lockstep_probe.py --stencil ssg --part 1 --size 512 --passes 4 --no-bits --shapes ${S2}_w2 ${S2}_lo_w2 ${S2}_lo_ls16_w2 ${S2}_lo_ls64_w2 ${S2}_ls64_w2 > $O/ls_ssg_p1_512.log 2>&1; tail -1 $O/ls_ssg_p1_512.log | cut -c1-900
timeout 1200 python3 -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -30 > $O/gpu_tests.txt; tail -6 $O/gpu_tests.txt
timeout 400 python3 bench.py > $O/bench_n1_default.json 2> $O/bench_n1_default.err; echo "bench rc=$?"
python3 - <<'PY'
import json,os
o=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r5h/bench_n1_default.json")
try:
    j=json.loads(open(o).read().strip().splitlines()[-1])
    print("value",j["value"],"ms",j["ms_per_step"],"frac",j["roofline"]["frac"],"placement",j["config"]["var_placement"])
except Exception as e: print("no bench line",e)
PY
timeout 600 python3 tools/generic_table.py --out $O --size3 512 --tag table512 --only iso3dfd 3axis 3axis_r1 ssg ssg2 ssg_merged awp awp_abc awp_elastic awp_elastic_abc tti iso3dfd_sponge test_3d test_boundary_3d test_stages_3d test_stream_3d test_scratch_3d test_partial_3d cube 3plane 3axis_with_diags fsg fsg2 fsg_abc fsg2_abc fsg_merged fsg_merged_abc > $O/table_512.log 2>&1; tail -4 $O/table_512.log
timeout 600 python3 tools/generic_table.py --out $O --size3 256 --tag table256 > $O/table_256.log 2>&1; tail -2 $O/table_256.log
