#!/bin/bash
# GPU job r6d: conditional scratch parts get bounding boxes (swe2d / wave2d), 4-D on the vector point kernel, 8-byte-lane / one-wave
# plane-ring shapes for parts with partial-dim tables: parity first, then what it buys.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6d; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
( time timeout 1500 python3 -m pytest tests/test_reference_stencils_gpu.py tests/test_multi_tile_fixtures_gpu.py tests/test_compile_time_variants_gpu.py tests/test_part_boxes_gpu.py tests/test_reference_api_programs_gpu.py tests/test_python_api_gpu.py -m gpu -q --timeout 600 -x 2>&1 | grep -v "^Solution '" ) > $O/parity.txt 2>&1
tail -n 25 $O/parity.txt
TWO="swe2d wave2d wave2d_f64 box_filter gaussian_filter test_2d test_boundary_2d test_scratch_2d test_stages_2d test_stream_2d test_4d test_scratch_3d test_scratch_boundary_1d"
python3 tools/generic_table.py --out $O --only $TWO test_partial_3d --size3 512 --tag after_default > $O/after.log 2>&1; cat $O/after.log
python3 tools/generic_table.py --out $O --only swe2d wave2d --size3 512 --opts "-hip_step_graphs 1" --tag after_graphs > $O/after_graphs.log 2>&1; cat $O/after_graphs.log
python3 - <<PY
import json
for tag in ("after_default",):
    for r in json.load(open("$O/%s.json" % tag)):
        if "parts" in r:
            fam = {}
            for p in r["parts"]:
                fam[p["kernel"]] = fam.get(p["kernel"], 0) + 1
            print(tag, r["stencil"], r["step_ms"], r["frac"], fam)
PY
python3 - <<PY
# every registered shape of test_partial_3d at 512^3, timed
import sys
sys.path.insert(0, "$R")
from yask_amd import yk_factory
from yask_amd.kernel import yk_env
yk_env.disable_debug_output()
fac = yk_factory("test_partial_3d")
s = fac.new_solution(fac.new_env())
s.set_overall_domain_size_vec([512, 512, 512])
s.prepare_solution()
for k, v in enumerate(s.get_vars()):
    v.set_elements_hash(1.0 + 0.25 * k, 0.1, hash_id=k)
print("chosen:", s.get_kernel_variant(0))
for i, n in enumerate(s.get_kernel_variant_names(0)):
    try:
        s.time_part(part=0, variant=i, t=0, reps=1)
        print("  %-44s %.3f ms  scratch %d B" % (n, s.time_part(part=0, variant=i, t=0, reps=5), s.get_kernel_variant_scratch_bytes(0, i)))
    except Exception as e:
        print("  ", n, "failed:", e)
PY
