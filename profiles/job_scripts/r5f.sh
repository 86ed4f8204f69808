#!/bin/bash
# GPU job r5f: lock-step every 16 / 32 / 64 planes (time, FETCH_SIZE), sweeps of every compiled shape for the 3-D solutions of the
# generic registry that sit below 0.5 of their HBM roofline, the default bench line with the hot placement trials (box 3).
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5f; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
timeout 400 python3 bench.py > $O/bench_n1_default.json 2> $O/bench_n1_default.err; echo "bench rc=$?"
python3 - <<'PY'
import json,os
o=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r5f/bench_n1_default.json")
try:
    j=json.loads(open(o).read().strip().splitlines()[-1])
    print("value",j["value"],"ms",j["ms_per_step"],"placement",j["config"]["var_placement"],"load",j["device_state"]["under_load"])
except Exception as e: print("no bench line",e)
PY
export YASK_HIP_LIB_DIR=$R/yask_amd/lib_prof
timeout 300 python3 tools/lockstep_probe.py --passes 4 --only 8 16 32 64 > $O/lockstep.log 2>&1; tail -3 $O/lockstep.log
B=starlin_v2_z128_y32_r4_m_nt_w2_c4
cd /tmp
for sh in $B ${B/_nt_/_nt_ls32_} ${B/_nt_/_nt_ls64_}; do
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/pmc_${sh}_FETCH -- python3 $R/tools/lockstep_probe.py --fetch $sh > $O/pmc_${sh}_FETCH.log 2>&1
done
cd $R
unset YASK_HIP_LIB_DIR
python3 - <<'PY'
import csv, glob, collections, os, json
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r5f")
res = {}
for d in sorted(glob.glob(O + "/pmc_*")):
    if not os.path.isdir(d): continue
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "starlin" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    res[os.path.basename(d)] = {k: [len(v), sum(v) / len(v)] for k, v in acc.items()}
json.dump(res, open(O + "/pmc_summary.json", "w"), indent=1); print(res)
PY
rm -rf $O/pmc_*/ 2>/dev/null
for spec in "test_3d 0 8" "awp 0 52" "awp_elastic 1 68" "tti 0 56" "iso3dfd_sponge 0 16" "awp 1 144" "test_boundary_3d 0 12"; do
  set -- $spec
  timeout 240 python3 tools/sweep_variants.py --stencil $1 --part $2 --size 512 --reps 5 --chunks 0 --bytes-per-point $3 --out $O/sweep_$1_p$2.json > $O/sweep_$1_p$2.log 2>&1
  grep "BEST" $O/sweep_$1_p$2.log | cut -c1-600
done
