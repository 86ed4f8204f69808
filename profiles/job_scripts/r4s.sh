#!/bin/bash
# GPU job r4s: bench.py N=4 at the headline size on one GPU stalls in the first run of the main solution (r4r) -- without the self-check? with a named transport / schedule?
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4s; mkdir -p $O; cd $R
export YASK_DIST_BACKEND=gloo YASK_BENCH_STACK_DUMP_S=45 YASK_HIP_WAIT_TIMEOUT_S=3 YASK_HIP_IPC_VERBOSE=1
run() { tag=$1; shift
  timeout 60 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port $((29950 + RANDOM % 40)) bench.py --gpus 4 --steps 6 --warmup 2 --ramp-secs 0.2 --no-probe "$@" > $O/$tag.json 2> $O/$tag.err
  echo "== $tag rc=$?"; python - <<PY
import json
try:
    j = json.loads([l for l in open("$O/$tag.json") if l.startswith("{")][0]); c = j["config"]
    print(j["value"], j["ms_per_step"], c["halo_transport"], c["schedule"], c["schedule_trials_ms_per_step"], j["halo"]["ms_per_step"])
except Exception as e:
    print("no line"); print("\n".join(l for l in open("$O/$tag.err").read().splitlines() if l.strip() and not l.startswith(("Solution", "[Gloo]", "/opt/amdgpu", "[W9", "***", "Setting OMP")))[:2500])
PY
}
run ipc_planned_nocheck --transport ipc --schedule planned --no-self-check
run ipc_serial_nocheck --transport ipc --schedule serial --no-self-check
