#!/bin/bash
# GPU job r4u: after ordering the mutual hipIpcOpenMemHandle calls of a link: bench.py N=4 at the headline size (everything on auto), IPC tests
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4u; mkdir -p $O; cd $R
export YASK_DIST_BACKEND=gloo YASK_BENCH_STACK_DUMP_S=80
timeout 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29981 bench.py --gpus 4 --steps 6 --warmup 2 --ramp-secs 0.2 --no-probe > $O/bench_n4.json 2> $O/bench_n4.err
echo "rc=$?"; python - <<PY
import json
try:
    j = json.loads([l for l in open("$O/bench_n4.json") if l.startswith("{")][0]); c = j["config"]
    print(j["value"], j["ms_per_step"], c["decomposition"], c["halo_transport"], c["transport_trials_ms_per_step"], c["schedule"], c["schedule_trials_ms_per_step"])
    print({k: v["ok"] for k, v in c["self_check"]["transports"].items()}, j["halo"]["ipc_control_plane_rank0"])
except Exception as e:
    print("no line", e); print("\n".join(l for l in open("$O/bench_n4.err").read().splitlines() if l.strip() and not l.startswith(("Solution", "[Gloo]", "/opt/amdgpu", "[W9", "***", "Setting OMP")))[:2000])
PY
unset YASK_DIST_BACKEND YASK_BENCH_STACK_DUMP_S
timeout 200 python -m pytest tests/test_transport_gpu.py -m gpu -q -x -k "keeps_the_host or native_bootstrap or four_ranks or pipelined" > $O/transport.log 2>&1; grep -E "passed|failed" $O/transport.log | tail -1
