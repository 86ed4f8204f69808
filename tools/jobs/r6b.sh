#!/bin/bash
# GPU job r6b: (1) the whole GPU suite on the tree after the ADVICE r05 fixes; (2) XCD map A/B on time, alternating processes
# (FETCH_SIZE said: strips 13.785 GB, 4x2 blocks 13.789, 2x4 blocks 14.236 -- r6a); (3) the 512^3 instruments side by side:
# tools/generic_table.py with the round-5 protocol (12 steps, no ramp), with the hot ramp, and bench.py --size 512.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6b; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
( time timeout 1500 python3 -m pytest tests -m gpu -q --timeout 600 -x 2>&1 | grep -v "^Solution '" ) > $O/gpu_tests.txt 2>&1
tail -n 8 $O/gpu_tests.txt
V=starlin_v4_z128_y32_r2_t2_nt_pd2_tl_w2_c2
for rep in 1 2 3; do for m in 0 1 2; do
  YASK_HIP_LIB_DIR=$R/yask_amd/lib_prof YASK_HIP_XCD_MAP=$m timeout 300 python3 bench.py --no-cpu-baseline --no-probe --traffic none --steps 40 --warmup 5 --opts "-hip_variant $V" 2>/dev/null \
    | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('xcd_map', $m, 'rep', $rep, 'ms_per_step', j['ms_per_step'], 'kernel_ms', j['roofline']['kernel_ms'], 'placement_sets', j['config']['var_placement']['ms_per_step_of_each_set'])"
done; done | tee $O/xcd_map_ab.txt
python3 tools/generic_table.py --out $O --only iso3dfd 3axis ssg --size3 512 --steps 12 --ramp-secs 0 --tag table512_r5_protocol > $O/t1.log 2>&1; cat $O/t1.log
python3 tools/generic_table.py --out $O --only iso3dfd 3axis ssg --size3 512 --tag table512_hot_ramp > $O/t2.log 2>&1; cat $O/t2.log
for i in 1 2; do timeout 300 python3 bench.py --size 512 --no-cpu-baseline --steps 200 --warmup 5 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('bench --size 512: ms_per_step', j['ms_per_step'], 'kernel_ms', j['roofline']['kernel_ms'], 'frac', j['roofline']['frac'], 'traffic', j['roofline']['traffic'], j['config']['kernel'])"; done | tee $O/bench512.txt
timeout 300 python3 bench.py --size 512 --no-cpu-baseline --steps 200 --warmup 5 --opts "-hip_placement_trials 1" 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('bench --size 512 first-draw placement: ms_per_step', j['ms_per_step'], 'kernel_ms', j['roofline']['kernel_ms'])" | tee -a $O/bench512.txt
