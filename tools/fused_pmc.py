#!/usr/bin/env python
"""A few fused steps of a 2-D solution under rocprofv3 counters (run by tools/jobs/r6j.sh): python tools/fused_pmc.py swe2d 4096"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
os.environ.setdefault("YASK_HIP_FUSE_SCRATCH", "1")
from yask_amd import yk_factory          # noqa: E402
from yask_amd.kernel import yk_env       # noqa: E402

yk_env.disable_debug_output()
st, n = sys.argv[1], int(sys.argv[2])
fac = yk_factory(st)
s = fac.new_solution(fac.new_env())
s.set_overall_domain_size_vec([n, n])
s.apply_command_line_options("-no-auto_tune -hip_step_graphs 0")
s.prepare_solution()
for k, v in enumerate(s.get_vars()):
    v.set_elements_hash(1.0 + 0.25 * k, 0.1, hash_id=k)
s.run_solution(0, 9)
print("fused groups:", s.get_fused_groups())
