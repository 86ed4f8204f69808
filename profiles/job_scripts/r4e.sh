#!/bin/bash
# GPU job r4e: why does bench.py N=4 over gloo + ipc (one GPU) stall?  Short waiter timeout, stderr kept; with and without the self-check.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4e; mkdir -p $O; cd $R
export YASK_DIST_BACKEND=gloo YASK_HIP_WAIT_TIMEOUT_S=3 YASK_HIP_IPC_VERBOSE=1
run() { tag=$1; shift
  timeout 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port $((29711 + RANDOM % 200)) bench.py --gpus 4 --steps 5 --warmup 2 --size 256 --ramp-secs 0.2 --no-probe "$@" > $O/$tag.out 2> $O/$tag.err
  echo "== $tag rc=$?"; tail -c 600 $O/$tag.out; grep -v "^Solution\|^$\|amdgpu.ids\|socket.cpp\|Gloo" $O/$tag.err | head -60; }
run noselfcheck --transport ipc --schedule planned --no-self-check
run selfcheck --transport ipc --schedule planned
timeout 60 python -m pytest tests/test_reference_api_programs_gpu.py -m gpu -q 2>&1 | tail -5
timeout 100 python bench.py --steps 20 --no-cpu-baseline --traffic none 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['bandwidth_probe'])"
