#!/bin/bash
# GPU job r4p: 3axis fp64, halo vectors of the next plane requested late (_hl1 / _hl2) against with-the-interior (default), time + FETCH_SIZE
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4p; mkdir -p $O; cd $R
python - <<'PY' 2>&1 | tee $O/hl_probe.log
import json, sys
sys.path.insert(0, ".")
from yask_amd import yk_factory
from yask_amd.kernel import yk_env
yk_env.disable_debug_output()
fac = yk_factory("3axis")
for n, fam in ((1024, "starlin_v2_z128_y32_r4_m_nt"), (512, "starlin_v2_z64_y32_r2_u_nt")):
    s = fac.new_solution(fac.new_env()); s.set_overall_domain_size_vec([n, n, n]); s.apply_command_line_options("-no-auto_tune"); s.prepare_solution()
    for k, v in enumerate(s.get_vars()): v.set_elements_hash(1.0, 0.1, hash_id=k)
    names = s.get_kernel_variant_names(0)
    cand = [i for i, x in enumerate(names) if x.startswith(fam) and "_wt" not in x]
    for i in cand: s.time_part(0, i, 0, 0, 3)
    for p in range(3):
        print(n, {names[i]: round(s.time_part(0, i, 0, 0, 12 if n > 512 else 30), 4) for i in cand}, flush=True)
    s.end_solution()
PY
cd /tmp; export TMPDIR=/tmp
for v in starlin_v2_z128_y32_r4_m_nt_w2_c4 starlin_v2_z128_y32_r4_m_nt_hl1_w2_c4 starlin_v2_z128_y32_r4_m_nt_hl2_w2_c4; do
  timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/fetch_$v -- python $R/bench.py --workload 3axis --size 1024 --steps 6 --warmup 2 --ramp-secs 0 --no-cpu-baseline --no-probe --traffic none "--opts=-hip_placement_trials 1 -hip_variant $v" > $O/fetch_$v.log 2>&1
  python - <<PY
import csv, glob
f = []
for p in glob.glob("$O/fetch_$v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "starlin" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE": f.append(float(r["Counter_Value"]))
if f: print("$v", "launches", len(f), "fetch GB (x2-corrected)", round(sum(f) / len(f) * 2048e-9, 3), "= x", round(sum(f) / len(f) * 2048 / 8589934592, 4), "of the algorithmic reads")
PY
done
