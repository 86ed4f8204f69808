#!/bin/bash
# Runs on the GPU box: FETCH_SIZE / WRITE_SIZE / L2 hit counters for one kernel variant of a workload.
# usage: tools/gpu_fetch.sh <tag> <variant> [bench args...]
set -u
TAG=$1; VAR=$2; shift 2
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/fetch_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --no-cpu-baseline --steps 6 --warmup 2 --opts '-hip_variant $VAR' $*"
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $pass | tr ' ' '_')
  eval rocprofv3 --kernel-trace --pmc $pass -f csv -d $OUT/$name -- $CMD > $OUT/$name.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "starlin" in r["Kernel_Name"] or "star25d" in r["Kernel_Name"] or "march" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: sum(v) / len(v) for k, v in acc.items()}
if "FETCH_SIZE" in avg:
    print("$VAR", "fetch_GB(x2 corrected)=%.3f" % (avg["FETCH_SIZE"] * 1024 * 2 / 1e9), "write_GB=%.3f" % (avg.get("WRITE_SIZE", 0) * 1024 / 1e9),
          "l2hit=%.3f" % (avg.get("TCC_HIT_sum", 0) / max(1.0, avg.get("TCC_HIT_sum", 0) + avg.get("TCC_MISS_sum", 0))))
PY
