#!/bin/bash
# GPU job r6i: ring-shaped 2-D conditions as box-minus-hole (strips for the sweeps, two box tests in the fused kernel): parity, times.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6i; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
( time timeout 900 python3 -m pytest tests/test_fused_scratch_gpu.py tests/test_reference_stencils_gpu.py tests/test_part_boxes_gpu.py tests/test_compile_time_variants_gpu.py tests/test_reference_api_programs_gpu.py -m gpu -q --timeout 300 2>&1 | grep -v "^Solution '" ) > $O/tests.txt 2>&1
tail -n 12 $O/tests.txt
TWO="swe2d wave2d wave2d_f64 test_scratch_2d test_boundary_2d"
YASK_HIP_FUSE_SCRATCH=0 python3 tools/generic_table.py --out $O --only $TWO --tag unfused > $O/unfused.log 2>&1; cat $O/unfused.log
YASK_HIP_FUSE_SCRATCH=1 python3 tools/generic_table.py --out $O --only $TWO --tag fused > $O/fused.log 2>&1; cat $O/fused.log
python3 tools/generic_table.py --out $O --only $TWO --tag default > $O/default.log 2>&1; cat $O/default.log
python3 - <<PY
import json
r = {x["stencil"]: x for x in json.load(open("$O/unfused.json"))}["swe2d"]
fam = {}
for p in r["parts"]:
    fam[p["kernel"]] = fam.get(p["kernel"], 0) + 1
print("swe2d unfused kernels:", fam, "sum of parts", r["sum_part_ms"])
print("  parts over 0.03 ms:", [(p["name"], p["kernel"], p["ms"]) for p in r["parts"] if (p["ms"] or 0) > 0.03])
PY
