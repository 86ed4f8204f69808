#!/bin/bash
# GPU job r4i: (1) ssg 512^3 one rank / 8 ranks / 8 ranks with -Mbt 2 (config 5's temporal blocking) bit for bit; (2) where do 3axis
# fp64's unexplained 5 % of reads come from?  FETCH_SIZE of the 128x32 large-grid shape against the 64x32 shape at 1024^3 (half the
# bytes per plane and XCD in flight): if intra-XCD halo lines fall out of the L2, the smaller tile must fetch less; (3) the probe again
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4i; mkdir -p $O; cd $R
timeout 200 python -m pytest tests/test_decomposed_blocks_gpu.py -m gpu -q -k ssg_512 2>&1 | tail -8 | cut -c1-300
cd /tmp; export TMPDIR=/tmp
for v in starlin_v2_z128_y32_r4_m_nt_w2_c4 starlin_v2_z64_y32_r2_u_nt_tl_w2_c4 starlin_v2_z64_y32_r2_m_nt_w2_c4; do
  timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/fetch_$v -- python $R/bench.py --workload 3axis --size 1024 --steps 6 --warmup 2 --ramp-secs 0 --no-cpu-baseline --no-probe --traffic none "--opts=-hip_placement_trials 1 -hip_variant $v" > $O/fetch_$v.log 2>&1
  python - <<PY
import csv, glob
f, d = [], []
for p in glob.glob("$O/fetch_$v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "starlin" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE": f.append(float(r["Counter_Value"]))
for p in glob.glob("$O/fetch_$v/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "starlin" in r["Kernel_Name"]: d.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
if f: print("$v", "launches", len(f), "fetch GB (x2-corrected)", round(sum(f) / len(f) * 2048e-9, 3), "= x", round(sum(f) / len(f) * 2048 / 8589934592, 4), "of the algorithmic reads; ms", round(sum(d) / max(1, len(d)), 4))
PY
done
cd $R; timeout 120 python bench.py --steps 20 --no-cpu-baseline --traffic none 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['bandwidth_probe'], j['roofline']['frac_of_this_box_copy'])"
