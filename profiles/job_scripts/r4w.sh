#!/bin/bash
# GPU job r4w: the IPC transport no longer maps var storage (x faces packed): transport + decomposed-block tests, bench.py N=4 and N=2 at the headline size
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4w; mkdir -p $O; cd $R
timeout 200 python -m pytest tests/test_transport_gpu.py tests/test_decomposed_blocks_gpu.py -m gpu -q > $O/tests.log 2>&1; grep -E "passed|failed" $O/tests.log | tail -1; grep -E "^FAILED|^ERROR" $O/tests.log | head -5
export YASK_DIST_BACKEND=gloo YASK_BENCH_STACK_DUMP_S=70
for n in 4 2; do
  timeout 90 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29990 + n)) bench.py --gpus $n --steps 6 --warmup 2 --ramp-secs 0.2 --no-probe > $O/bench_n$n.json 2> $O/bench_n$n.err
  echo "n=$n rc=$?"; python - <<PY
import json
try:
    j = json.loads([l for l in open("$O/bench_n$n.json") if l.startswith("{")][0]); c = j["config"]
    print(j["value"], j["ms_per_step"], c["decomposition"], c["halo_transport"], c["transport_trials_ms_per_step"], c["schedule"], c["schedule_trials_ms_per_step"])
    print({k: v["ok"] for k, v in c["self_check"]["transports"].items()}, j["halo"]["ipc_control_plane_rank0"], j["halo"]["ms_per_step"])
except Exception as e:
    print("no line", e)
PY
done
