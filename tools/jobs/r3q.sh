#!/bin/bash
# GPU job r3q: the whole bench path with 8 and 2 ranks sharing one GPU (gloo rendezvous, transport + schedule auto), full gpu suite
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3q; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
for n in 8 2; do
echo "== bench $n ranks on one GPU (gloo, transport auto)"; ( time YASK_DIST_BACKEND=gloo timeout 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2972$n bench.py --gpus $n --steps 10 --warmup 2 --ramp-secs 0.3 --no-probe ) > $O/bench_n$n.log 2>&1; echo "rc=$?"; grep '^{' $O/bench_n$n.log > $O/bench_n$n.json; python -c "
import sys,json; j=json.loads(open('$O/bench_n$n.json').read()); print(j['value'], j['ms_per_step'], j['config']['halo_transport'], j['config']['transport_trials_ms_per_step'], j['config']['schedule'], j['config']['schedule_trials_ms_per_step'], j.get('halo'))" 2>&1 | tail -3; tail -2 $O/bench_n$n.log | cut -c1-300
done
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -n "passed\|failed" $O/pytest.log | tail -3
