#!/bin/bash
# GPU job r4h: the whole GPU suite on the round-4 tree; rocprofv3 evidence (kernel trace + PMC passes) for the headline, the 512^3
# block, 3axis fp64 at both sizes and ssg -- the post-5734517 kernels (r03l_* predate them); the default bench line
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4h; mkdir -p $O; cd $R
timeout 600 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -60 > $O/suite.log; tail -22 $O/suite.log | cut -c1-300
prof() { tag=$1; shift; timeout 400 python tools/gpu_profile.py $tag --pmc-steps 6 -- --traffic none --no-probe "$@" > $O/prof_$tag.log 2>&1; python - <<PY
import json
try:
    s = json.load(open("$R/gpurun_out/prof_$tag/summary.json"))
    print("$tag", {k: s.get(k) for k in ("sum_of_hot_kernel_avg_ms", "roofline_frac_at_rocprof_duration", "traffic_over_algorithmic")}, s["bench_line_of_the_profiled_run"]["ms_per_step"])
except Exception as e:
    print("$tag: no summary", e)
PY
}
prof r4_iso3dfd
prof r4_iso512 --size 512
prof r4_3axis1024 --workload 3axis --size 1024
prof r4_3axis512 --workload 3axis
prof r4_ssg --workload ssg
timeout 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-1200 $O/bench_default.json; python -c "
import json; j=json.load(open('$O/bench_default.json')); print(j['bandwidth_probe']); print(j['cpu_baseline']); print(j['roofline'])"
