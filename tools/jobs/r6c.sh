#!/bin/bash
# GPU job r6c: (1) parity of the round's new kernel paths -- 2-D parts lifted to the 3-D families (ykh_lift2d.hpp), partial-dim tables
# on the plane-ring kernel (kind 4), the "_np" twins of the no-packed translation unit -- against the reference goldens (one-tile,
# the new 2-D multi-tile ones, the 3-D multi-tile lattices, the compile-time variants); (2) what they buy: generic_table on the
# 2-D solutions and test_partial_3d, forced point kernel vs the timed default; the solutions with _np twins.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6c; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
( time timeout 1500 python3 -m pytest tests/test_reference_stencils_gpu.py tests/test_multi_tile_fixtures_gpu.py tests/test_compile_time_variants_gpu.py tests/test_box_kernel_gpu.py tests/test_clusters_gpu.py tests/test_part_boxes_gpu.py -m gpu -q --timeout 600 2>&1 | grep -v "^Solution '" ) > $O/parity.txt 2>&1
tail -n 25 $O/parity.txt
TWO="swe2d wave2d wave2d_f64 box_filter gaussian_filter test_2d test_boundary_2d test_scratch_2d test_stages_2d test_stream_2d"
python3 tools/generic_table.py --out $O --only $TWO test_partial_3d --size3 512 --opts "-hip_variant naive" --tag before_naive > $O/before.log 2>&1; cat $O/before.log
python3 tools/generic_table.py --out $O --only $TWO test_partial_3d --size3 512 --tag after_default > $O/after.log 2>&1; cat $O/after.log
python3 tools/generic_table.py --out $O --only swe2d wave2d --size3 512 --opts "-hip_step_graphs 1" --tag after_graphs > $O/after_graphs.log 2>&1; cat $O/after_graphs.log
python3 tools/generic_table.py --out $O --only cube 3plane 3axis_with_diags tti test_scratch_3d awp_abc awp_elastic_abc --size3 512 --tag np_twins > $O/np.log 2>&1; cat $O/np.log
python3 - <<PY
import json
for tag in ("after_default", "np_twins"):
    for r in json.load(open("$O/%s.json" % tag)):
        if "parts" in r:
            print(tag, r["stencil"], r["step_ms"], r["frac"], sorted({p["kernel"] for p in r["parts"]})[:6])
PY
