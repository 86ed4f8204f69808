#!/bin/bash
# GPU job r4o: write-through (sc1) output stores against the non-temporal ones, same allocations, alternating (tools/wt_probe.py);
# FETCH_SIZE of the 3axis large-grid shape with and without
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4o; mkdir -p $O; cd $R
timeout 400 python tools/wt_probe.py 3 2>&1 | tee $O/wt_probe.log | cut -c1-400
cp gpurun_out/wt_probe.json $O/ 2>/dev/null
cd /tmp; export TMPDIR=/tmp
for v in starlin_v2_z128_y32_r4_m_nt_w2_c4 starlin_v2_z128_y32_r4_m_nt_wt_w2_c4; do
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
