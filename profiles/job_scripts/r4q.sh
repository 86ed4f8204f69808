#!/bin/bash
# GPU job r4q: last full run of the GPU suite + smoke on the final tree (kernel headers gained the _wt / halo-late code paths, compiled out)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4q; mkdir -p $O; cd $R
timeout 600 python -m pytest tests -m gpu -q --durations=5 > $O/suite.log 2>&1; grep -E "passed|failed" $O/suite.log | tail -2; grep -E "^FAILED|^ERROR" $O/suite.log | head
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['traffic'], j['roofline']['frac_of_this_box_copy'])"
