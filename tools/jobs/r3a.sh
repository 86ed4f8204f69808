#!/bin/bash
# GPU job r3a (round 3, first call): new GPU tests (fuse_vars, IPC transport, planned launches, decomposed blocks at the
# BASELINE sizes), the compute-side cost of the planned launches (tools/decomp_cost.py), the headline bench (regression check
# after the kernel-prologue change), and overlap evidence of the IPC transport with 8 ranks sharing the GPU.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3a
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== quick new tests"; ( time timeout 600 python -m pytest tests/test_fuse_vars_gpu.py tests/test_transport_gpu.py tests/test_fused_gpu.py tests/test_placement_gpu.py -q --timeout 240 --durations=15 ) > $O/pytest_new.log 2>&1; echo "rc=$?"; tail -25 $O/pytest_new.log
echo "== decomposed blocks at BASELINE sizes"; ( time timeout 900 python -m pytest tests/test_decomposed_blocks_gpu.py -q -s --timeout 400 --durations=10 ) > $O/pytest_blocks.log 2>&1; echo "rc=$?"; tail -30 $O/pytest_blocks.log
echo "== decomp cost iso3dfd"; ( time timeout 420 python tools/decomp_cost.py --stencil iso3dfd ) > $O/decomp_iso3dfd.log 2>&1; echo "rc=$?"; cp gpurun_out/decomp_cost_iso3dfd.json $O/ 2>/dev/null; tail -42 $O/decomp_iso3dfd.log
echo "== decomp cost ssg"; ( time timeout 300 python tools/decomp_cost.py --stencil ssg ) > $O/decomp_ssg.log 2>&1; echo "rc=$?"; cp gpurun_out/decomp_cost_ssg.json $O/ 2>/dev/null; tail -18 $O/decomp_ssg.log
echo "== headline bench"; ( time timeout 300 python bench.py --no-cpu-baseline ) > $O/bench_n1.log 2>&1; echo "rc=$?"; tail -2 $O/bench_n1.log | cut -c1-1500
echo "== 8 ranks on one GPU, ipc"; ( time YASK_DIST_BACKEND=gloo timeout 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 8 --steps 20 --warmup 3 --transport ipc --ramp-secs 0.5 --no-probe ) > $O/bench_n8_ipc.log 2>&1; echo "rc=$?"; grep '^{' $O/bench_n8_ipc.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['config']['schedule'], j['config']['schedule_trials_ms_per_step'], j['halo'])" 2>&1 | tail -3
echo "== rest of the gpu suite"; ( time timeout 900 python -m pytest tests -m gpu -q --timeout 300 --durations=15 --deselect tests/test_decomposed_blocks_gpu.py --deselect tests/test_transport_gpu.py --deselect tests/test_fuse_vars_gpu.py --deselect tests/test_fused_gpu.py --deselect tests/test_placement_gpu.py ) > $O/pytest_rest.log 2>&1; echo "rc=$?"; tail -25 $O/pytest_rest.log
