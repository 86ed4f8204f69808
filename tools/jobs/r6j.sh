#!/bin/bash
# GPU job r6j: the fused scratch kernel after the v_min3 form of its per-point tests: step times, and where its cycles go (SQ counters).
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6j; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
timeout 600 python3 -m pytest tests/test_fused_scratch_gpu.py -m gpu -q --timeout 300 2>&1 | tail -n 3
YASK_HIP_FUSE_SCRATCH=1 python3 tools/generic_table.py --out $O --only swe2d wave2d wave2d_f64 test_scratch_2d --tag fused > $O/fused.log 2>&1; cat $O/fused.log
cd /tmp
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_IFETCH SQ_WAVES" "GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-30)
  timeout 300 rocprofv3 --kernel-trace --pmc $pass -f csv -d $O/pmc_$name -- python3 $R/tools/fused_pmc.py swe2d 4096 > $O/pmc_$name.log 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$O/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fused2d" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-28s %.4g  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
rm -rf $O/pmc_*/
