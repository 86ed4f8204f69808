#!/bin/bash
# GPU job r4r: bench.py at the headline size with 2 and 4 ranks sharing the one GPU (gloo process group, IPC / host-staged halos): the flow the
# driver's scaling runs take (self-check, transport and schedule trials, timed region, JSON line), at real message sizes
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4r; mkdir -p $O; cd $R
export YASK_DIST_BACKEND=gloo YASK_BENCH_STACK_DUMP_S=140
for n in 2 4; do
  timeout 170 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29900 + n)) bench.py --gpus $n --steps 10 --warmup 3 --ramp-secs 0.5 --no-probe > $O/bench_n$n.json 2> $O/bench_n$n.err
  echo "n=$n rc=$?"; python - <<PY
import json
try:
    j = json.loads([l for l in open("$O/bench_n$n.json") if l.startswith("{")][0])
    c = j["config"]
    print(j["value"], "Gpoints/s", j["ms_per_step"], "ms/step;", c["workload"], "|", c["decomposition"], "| transport", c["halo_transport"], c["transport_trials_ms_per_step"], "| schedule", c["schedule"], c["schedule_trials_ms_per_step"])
    print("self_check", {k: (v["ok"], v["schedules"]) for k, v in c["self_check"]["transports"].items()}, "devices", c["self_check"]["devices"])
    print("halo", j["halo"]["bytes_sent_per_step_rank0"], j["halo"]["ms_per_step"], j["halo"].get("ipc_control_plane_rank0"))
except Exception as e:
    print("no line:", e); print(open("$O/bench_n$n.err").read()[-1500:])
PY
done
