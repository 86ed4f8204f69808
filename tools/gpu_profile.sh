#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + PMC passes for the bench command.
# usage: tools/gpu_profile.sh <tag> [bench args...]
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --no-cpu-baseline --steps 10 --warmup 2 $*"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $pass -f csv -d $OUT/pmc_$name -- $CMD > $OUT/pmc_$name.log 2>&1
done
# summarise: per-kernel average of each counter
python - <<PY
import csv, glob, collections, json, os
out = {}
for f in glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        for c, v in d.items():
            out.setdefault(k, {})[c] = {"avg": sum(v) / len(v), "n": len(v)}
json.dump(out, open("$OUT/pmc_summary.json", "w"), indent=1)
for k, d in out.items():
    if any(t in k for t in ("star25d", "starlin", "march", "vecpt", "naive")):
        print(k, {c: round(x["avg"], 1) for c, x in d.items()})
for f in glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:1500])
PY
