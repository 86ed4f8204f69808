#!/bin/bash
# GPU job r5g: the XCD lock-step on the HEADLINE kernel (iso3dfd 1024^3 and 512^3, profiling library), 3axis with its new large-grid
# default (bench line at 1024^3 + the 3axis parity tests), the 3-D part of the generic table after the tuner fix, bench box 4.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5g; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
YASK_HIP_LIB_DIR=$R/yask_amd/lib_prof timeout 300 python3 tools/lockstep_probe.py --stencil iso3dfd --passes 4 --only 16 64 > $O/lockstep_iso3dfd_1024.log 2>&1; tail -2 $O/lockstep_iso3dfd_1024.log
YASK_HIP_LIB_DIR=$R/yask_amd/lib_prof timeout 300 python3 tools/lockstep_probe.py --stencil iso3dfd --size 512 --passes 4 --only 16 64 > $O/lockstep_iso3dfd_512.log 2>&1; tail -2 $O/lockstep_iso3dfd_512.log | head -1
timeout 300 python3 bench.py --workload 3axis --size 1024 --no-cpu-baseline > $O/bench_3axis_1024.json 2> $O/bench_3axis_1024.err; echo "3axis bench rc=$?"
timeout 600 python3 -m pytest tests/test_stencils_gpu.py tests/test_baseline_configs_gpu.py tests/test_big_fixtures_gpu.py tests/test_fused_gpu.py -m gpu -q -k "3axis or heat or axis" --timeout 600 2>&1 | tail -5 > $O/tests_3axis.txt; cat $O/tests_3axis.txt
timeout 400 python3 bench.py > $O/bench_n1_default.json 2> $O/bench_n1_default.err; echo "bench rc=$?"
python3 - <<'PY'
import json,os
for f in ("bench_3axis_1024.json","bench_n1_default.json"):
    o=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r5g",f)
    try:
        j=json.loads(open(o).read().strip().splitlines()[-1])
        print(f,"value",j["value"],"ms",j["ms_per_step"],"frac",j["roofline"]["frac"],"traffic",j["roofline"]["traffic"],"kernel",j["config"]["kernel"],"placement",j["config"]["var_placement"])
    except Exception as e: print(f,"no bench line",e)
PY
timeout 600 python3 tools/generic_table.py --out $O --size3 512 --tag table512 --only iso3dfd 3axis 3axis_r1 ssg ssg2 ssg_merged awp awp_abc awp_elastic awp_elastic_abc tti iso3dfd_sponge test_3d test_boundary_3d test_stages_3d test_stream_3d test_scratch_3d test_partial_3d cube 3plane 3axis_with_diags fsg fsg2 > $O/table_512.log 2>&1; tail -25 $O/table_512.log
