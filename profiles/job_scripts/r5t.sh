#!/bin/bash
# GPU job r5t (= r5q on the final tree: + box lists of sub-domain conditions, narrow point-kernel tile): new tests first, iso3dfd_sponge
# shapes at 512^3, the full GPU suite, the generic tables (512^3 and 256^3), the default bench line.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5t; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
timeout 600 python3 -m pytest tests/test_clusters_gpu.py tests/test_box_kernel_gpu.py tests/test_part_boxes_gpu.py -m gpu -q --timeout 500 2>&1 | tail -12 > $O/new_tests.txt; tail -4 $O/new_tests.txt
timeout 300 python3 tools/sweep_variants.py --stencil iso3dfd_sponge --size 512 --reps 10 --chunks 0 --check --steps 20 --out $O/sweep_iso3dfd_sponge_p0.json > $O/sweep_sponge.log 2>&1; grep -E "WHOLE|FAILED|mismatches vs naive: [1-9]" $O/sweep_sponge.log | cut -c1-300; tail -1 $O/sweep_sponge.log | cut -c1-700
timeout 1500 python3 -m pytest tests -m gpu -q --timeout 900 --durations=15 2>&1 | tail -40 > $O/gpu_tests.txt; tail -5 $O/gpu_tests.txt
timeout 700 python3 tools/generic_table.py --out $O --size3 512 --tag table512 --only iso3dfd 3axis 3axis_r1 ssg ssg2 ssg_merged awp awp_abc awp_elastic awp_elastic_abc tti iso3dfd_sponge test_3d test_boundary_3d test_stages_3d test_stream_3d test_scratch_3d test_partial_3d cube 3plane 3axis_with_diags fsg fsg2 fsg_abc fsg2_abc fsg_merged fsg_merged_abc > $O/table_512.log 2>&1; tail -3 $O/table_512.log | cut -c1-300
timeout 500 python3 tools/generic_table.py --out $O --size3 256 --tag table256 > $O/table_256.log 2>&1; tail -2 $O/table_256.log | cut -c1-300
timeout 400 python3 bench.py > $O/bench_n1_default.json 2> $O/bench_n1_default.err; echo "bench rc=$?"
python3 - <<'PY'
import json,os
o=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r5t/bench_n1_default.json")
try:
    j=json.loads(open(o).read().strip().splitlines()[-1])
    print("value",j["value"],"ms",j["ms_per_step"],"frac",j["roofline"]["frac"],"cpu",j["cpu_baseline"]["value"])
except Exception as e: print("no bench line",e)
PY
