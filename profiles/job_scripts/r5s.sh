#!/bin/bash
# GPU job r5s: full boxes of sub-domain conditions (find_part_boxes): geometry tests, the golden parity of the solutions with conditions,
# then the *_abc solutions and test_boundary_3d at 512^3 (generic table: step time, per-part kernels).
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5s; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
timeout 600 python3 -m pytest tests/test_part_boxes_gpu.py -m gpu -q --timeout 500 2>&1 | tail -30 > $O/box_tests.txt; tail -25 $O/box_tests.txt
timeout 900 python3 -m pytest tests/test_reference_stencils_gpu.py tests/test_transport_gpu.py -m gpu -q --timeout 800 -k "abc or boundary or sub_domain or step_cond or picks_fast" 2>&1 | tail -8 > $O/parity.txt; tail -4 $O/parity.txt
timeout 700 python3 tools/generic_table.py --out $O --size3 512 --tag table512_abc --only awp_abc awp_elastic_abc fsg_abc fsg2_abc fsg_merged_abc test_boundary_3d > $O/table.log 2>&1; tail -8 $O/table.log | cut -c1-200
python3 - <<'PY'
import json,os
o=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r5s/table512_abc.json")
for r in json.load(open(o)):
    print(r['stencil'], 'step', r.get('step_ms'), 'frac', r.get('frac'))
    for p in r.get('parts', []): print('    ', p.get('name'), p.get('kernel'), 'ms', p.get('ms'), 'points', p.get('points'))
PY
