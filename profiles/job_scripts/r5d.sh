#!/bin/bash
# GPU job r5d: (1) the 3axis fp64 lock-step experiment (tools/lockstep_probe.py, profiling library) with FETCH_SIZE of the base and
# the _ls shapes; (2) HBM traffic of fsg's two vecpt kernels (why 0.16?); (3) the table again with reads per point.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5d; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
export YASK_HIP_LIB_DIR=$R/yask_amd/lib_prof
timeout 300 python3 tools/lockstep_probe.py --passes 3 > $O/lockstep.log 2>&1; tail -5 $O/lockstep.log
B=starlin_v2_z128_y32_r4_m_nt_w2_c4
cd /tmp
for sh in $B ${B/_nt_/_nt_ls2_} ${B/_nt_/_nt_ls4_} ${B/_nt_/_nt_ls8_}; do
  for c in FETCH_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    n=$(echo $c | cut -d' ' -f1)
    timeout 200 rocprofv3 --kernel-trace --pmc $c -f csv -d $O/pmc_${sh}_$n -- python3 $R/tools/lockstep_probe.py --fetch $sh > $O/pmc_${sh}_$n.log 2>&1
  done
done
unset YASK_HIP_LIB_DIR
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c -f csv -d $O/pmc_fsg_$c -- python3 -m yask_amd.harness -stencil fsg -g 256 -trial_steps 3 -num_trials 1 > $O/pmc_fsg_$c.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections, os, json
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r5d")
res = {}
for d in sorted(glob.glob(O + "/pmc_*")):
    if not os.path.isdir(d): continue
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if any(x in k for x in ("starlin", "vecpt", "march")):
                acc[(k.split("<")[0][-20:] + "<" + k.split("<")[1][:40] if "<" in k else k[:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    res[os.path.basename(d)] = {f"{k[0]} {k[1]}": [len(v), sum(v) / len(v)] for k, v in acc.items()}
json.dump(res, open(O + "/pmc_summary.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])
PY
rm -rf $O/pmc_*/ 2>/dev/null
timeout 600 python3 tools/generic_table.py --out $O --size3 512 --tag table512 > $O/table_512.log 2>&1; tail -5 $O/table_512.log
