#!/bin/bash
# GPU job r4v: N=4 full size, IPC, packed x faces (-no-hip_direct_halo: no var storage is mapped by a peer), control plane traced
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4v; mkdir -p $O; cd $R
export YASK_DIST_BACKEND=gloo YASK_HIP_WAIT_TIMEOUT_S=3 YASK_HIP_IPC_VERBOSE=2
timeout 50 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29983 bench.py --gpus 4 --steps 6 --warmup 2 --ramp-secs 0.2 --no-probe --transport ipc --schedule serial --no-self-check "--opts=-no-hip_direct_halo" > $O/out.json 2> $O/err.log
echo "rc=$?"; cut -c1-300 $O/out.json | head -2; grep -E "^ipc\[|waited in vain|<--" $O/err.log | tail -40 | cut -c1-160
