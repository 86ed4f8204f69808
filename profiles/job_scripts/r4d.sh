#!/bin/bash
# GPU job r4d: the whole suite after the variant collapse / the split of ykh_solution.cpp / the C++ API linked into the stencil
# libraries; then bench.py N=4 on one GPU over gloo (the dry-run case that failed in r4c), output kept
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4d; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q --durations=12 2>&1 | tail -120 > $O/suite.log; tail -40 $O/suite.log
export YASK_DIST_BACKEND=gloo YASK_HIP_WAIT_TIMEOUT_S=15
for n in 4; do
  timeout 170 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus $n --steps 5 --warmup 2 --size 256 --ramp-secs 0.2 > $O/bench_n$n.out 2> $O/bench_n$n.err
  echo "bench n=$n rc=$?"; tail -c 1500 $O/bench_n$n.out; grep -v "^Solution\|^$" $O/bench_n$n.err | tail -40
done
