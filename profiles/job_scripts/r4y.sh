#!/bin/bash
# GPU job r4y (the round's last seconds): bench.py N=2 on one GPU, everything on auto, with the IPC pre-flight in child processes
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4y; mkdir -p $O; cd $R
export YASK_DIST_BACKEND=gloo
timeout 70 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 4 --warmup 1 --size 256 --ramp-secs 0.1 --no-probe > $O/bench_n2.json 2> $O/bench_n2.err
echo "rc=$?"; python - <<PY
import json
try:
    j = json.loads([l for l in open("$O/bench_n2.json") if l.startswith("{")][0]); c = j["config"]
    print(j["value"], c["halo_transport"], c["transport_trials_ms_per_step"], "preflight", c["ipc_preflight_in_child_processes"], {k: v["ok"] for k, v in c["self_check"]["transports"].items()})
except Exception as e:
    print("no line", e); print("\n".join(l for l in open("$O/bench_n2.err").read().splitlines() if l.strip() and not l.startswith(("Solution", "[Gloo]", "/opt/amdgpu", "[W9", "***", "Setting OMP")))[-1800:])
PY
