#!/bin/bash
# GPU job r4t: the N=4 full-size stall with the IPC control plane traced (YASK_HIP_IPC_VERBOSE=2)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4t; mkdir -p $O; cd $R
export YASK_DIST_BACKEND=gloo YASK_HIP_WAIT_TIMEOUT_S=3 YASK_HIP_IPC_VERBOSE=2
timeout 45 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29977 bench.py --gpus 4 --steps 6 --warmup 2 --ramp-secs 0.2 --no-probe --transport ipc --schedule serial --no-self-check > $O/out.json 2> $O/err.log
echo "rc=$?"; grep -E "^ipc\[|yask ipc|waited in vain|<--|mailbox" $O/err.log | tail -70 | cut -c1-200
