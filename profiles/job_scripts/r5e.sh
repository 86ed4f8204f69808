#!/bin/bash
# GPU job r5e: the full GPU suite on the reduced surface (12 -hip_* options, four schedules, halves the default for decomposed runs),
# the default bench line (box 2 of the kept-vs-timed check), the lock-step experiment with the fixed poll, fsg's HBM traffic.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5e; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
timeout 1200 python3 -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -40 > $O/gpu_tests.txt; tail -12 $O/gpu_tests.txt
timeout 400 python3 bench.py > $O/bench_n1_default.json 2> $O/bench_n1_default.err; echo "bench rc=$?"
python3 - <<'PY'
import json,os
o=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r5e/bench_n1_default.json")
try:
    j=json.loads(open(o).read().strip().splitlines()[-1])
    print("value",j["value"],"ms",j["ms_per_step"],"placement",j["config"]["var_placement"],"load",j["device_state"]["under_load"])
except Exception as e: print("no bench line",e)
PY
export YASK_HIP_LIB_DIR=$R/yask_amd/lib_prof
timeout 300 python3 tools/lockstep_probe.py --passes 3 > $O/lockstep.log 2>&1; tail -3 $O/lockstep.log
B=starlin_v2_z128_y32_r4_m_nt_w2_c4
cd /tmp
for sh in $B ${B/_nt_/_nt_ls1_} ${B/_nt_/_nt_ls4_} ${B/_nt_/_nt_ls16_}; do
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/pmc_${sh}_FETCH -- python3 $R/tools/lockstep_probe.py --fetch $sh > $O/pmc_${sh}_FETCH.log 2>&1
done
unset YASK_HIP_LIB_DIR
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c -f csv -d $O/pmc_fsg_$c -- python3 -m yask_amd.harness -stencil fsg -g 256 -trial_steps 3 -num_trials 1 > $O/pmc_fsg_$c.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections, os, json
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r5e")
res = {}
for d in sorted(glob.glob(O + "/pmc_*")):
    if not os.path.isdir(d): continue
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if any(x in k for x in ("starlin", "vecpt", "march")):
                acc[(k[:110], r["Counter_Name"])].append(float(r["Counter_Value"]))
    res[os.path.basename(d)] = {f"{k[0]} {k[1]}": [len(v), sum(v) / len(v)] for k, v in acc.items()}
json.dump(res, open(O + "/pmc_summary.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:5000])
PY
rm -rf $O/pmc_*/ 2>/dev/null
