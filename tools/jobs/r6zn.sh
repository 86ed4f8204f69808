#!/bin/bash
# GPU job r6zn: the 3-D kernel families on a solution with four domain dims (test_4d): parity of every registered shape, table, shapes.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6zn; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
( timeout 600 python3 -m pytest tests/test_reference_stencils_gpu.py -m gpu -x -q --timeout 600 -k "test_4d or four_domain" 2>&1 | grep -v "^Solution '" ) > $O/parity.txt 2>&1
tail -n 3 $O/parity.txt
python3 tools/generic_table.py --out $O --only test_4d --tag t4d 2>&1
python3 - <<'PY'
import sys; sys.path.insert(0, '.')
from yask_amd import yk_factory
fac = yk_factory("test_4d"); env = fac.new_env()
s = fac.new_solution(env); s.set_overall_domain_size_vec([16, 256, 256, 256]); s.prepare_solution()
for k, v in enumerate(s.get_vars()): v.set_elements_hash(1.0 + 0.25 * k, 0.1, hash_id=k)
for vi, n in enumerate(s.get_kernel_variant_names(0)):
    print(n, round(s.time_part(0, vi, 0, 0, 3), 4))
PY
