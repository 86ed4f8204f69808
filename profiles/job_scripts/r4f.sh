#!/bin/bash
# GPU job r4f: bench.py N=4 over gloo, everything on auto (the r4c/r4d stall): stack of every rank after 40 s
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4f; mkdir -p $O; cd $R
export YASK_DIST_BACKEND=gloo YASK_HIP_WAIT_TIMEOUT_S=3 YASK_HIP_IPC_VERBOSE=1 YASK_BENCH_STACK_DUMP_S=40
timeout 75 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29833 bench.py --gpus 4 --steps 5 --warmup 2 --size 256 --ramp-secs 0.2 --no-probe > $O/auto.out 2> $O/auto.err
echo "rc=$?"; tail -c 1500 $O/auto.out; grep -v "^Solution\|^$\|amdgpu.ids\|socket.cpp\|Gloo\|^\*\*\*\|OMP_NUM" $O/auto.err | head -150
