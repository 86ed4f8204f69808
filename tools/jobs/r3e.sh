#!/bin/bash
# GPU job r3e: the whole -m gpu suite (as the driver runs it), smoke(), and the 8-ranks-on-one-GPU bench with transport auto-selection.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3e
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== gpu suite"; ( time timeout 1200 python -m pytest tests/ -x -q -m gpu --durations=12 ) > $O/pytest_all.log 2>&1; echo "rc=$?"; tail -22 $O/pytest_all.log
echo "== smoke"; ( time timeout 120 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "rc=$?"; tail -2 $O/smoke.log
echo "== bench 8 ranks on one GPU (gloo, transport auto)"; ( time YASK_DIST_BACKEND=gloo timeout 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29721 bench.py --gpus 8 --steps 10 --warmup 2 --ramp-secs 0.3 --no-probe ) > $O/bench_n8.log 2>&1; echo "rc=$?"; grep '^{' $O/bench_n8.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['config']['halo_transport'], j['config']['transport_trials_ms_per_step'], j['config']['schedule'], j['config']['schedule_trials_ms_per_step'], j['halo'])" 2>&1 | tail -3; tail -3 $O/bench_n8.log | cut -c1-300
