#!/bin/bash
# GPU job r5b: the full GPU suite on round 5's first library changes (guarded hipIpcOpenMemHandle, init guard, async error checks,
# placement trial = mean of 6 steps), the default bench line (sampler by PCI bus id, kept-vs-timed), then the stand-alone
# hipIpcOpenMemHandle probe (tools/microbench/ipc_open.hip) -- last, because a hang there is the thing looked for.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5b; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python3 -m pytest tests -m gpu -x -q --timeout 600 2>&1 | tail -15 > $O/gpu_tests.txt; tail -5 $O/gpu_tests.txt
timeout 400 python3 bench.py > $O/bench_n1_default.json 2> $O/bench_n1_default.err; echo "bench rc=$?"; head -c 600 $O/bench_n1_default.json; echo
python3 - <<'PY'
import json,os
o=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r5b/bench_n1_default.json")
try:
    j=json.loads(open(o).read().strip().splitlines()[-1])
    print("value",j["value"],"ms",j["ms_per_step"],"placement",j["config"]["var_placement"],"load",j["device_state"]["under_load"])
except Exception as e: print("no bench line",e)
PY
cd tools/microbench && hipcc -O3 --offload-arch=gfx950 ipc_open.hip -o /tmp/ipc_open -lpthread 2>&1 | grep -v warning | tail -3
hangs=0
for cfg in "2 3 0" "4 3 0" "8 3 0" "4 5 0" "2 3 1" "4 3 1" "4 3 2" "4 3 3" "8 3 3" "8 1 3"; do
  set -- $cfg
  timeout 120 /tmp/ipc_open $1 $2 $3 40 >> $O/ipc_open.txt 2>&1; rc=$?
  echo "-> rc=$rc" >> $O/ipc_open.txt
  [ $rc -eq 2 ] && hangs=$((hangs+1))
  [ $hangs -ge 2 ] && { echo "two hangs: stopping" >> $O/ipc_open.txt; break; }
done
cat $O/ipc_open.txt | tail -80
