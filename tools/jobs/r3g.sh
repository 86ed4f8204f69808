#!/bin/bash
# GPU job r3g: whole gpu suite after the _tl shapes and the 3axis additions; bench lines of ssg / 3axis / iso3dfd 512^3; rocprofv3 kernel
# trace of a decomposed rank's steps (planned launches, pack / unpack, waiters) through the mirror transport.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3g
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== gpu suite"; ( time timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=6 ) > $O/pytest_all.log 2>&1; echo "rc=$?"; tail -12 $O/pytest_all.log
for w in "ssg" "ssg --size 1024" "3axis" "3axis --size 1024" "iso3dfd --size 512" "heat3d --opts=-hip_fuse_steps_2"; do
  tag=$(echo $w | tr ' =' '__' | tr -d '-'); a=$(echo $w | sed 's/-hip_fuse_steps_2/-hip_fuse_steps 2/')
  timeout 300 python bench.py --workload $a --no-cpu-baseline --traffic none > $O/bench_$tag.log 2>&1
  grep '^{' $O/bench_$tag.log > $O/bench_$tag.json; python -c "
import json; j=json.load(open('$O/bench_$tag.json')); print('$w', j['value'], j['ms_per_step'], j['roofline']['frac'], j['config']['kernel'])"
done
echo "== rocprofv3 kernel trace of a decomposed rank (mirror transport)"
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_decomp -- python $R/tools/overlap_probe.py --stencil iso3dfd --steps 30 ) > $O/prof_decomp.log 2>&1; echo "rc=$?"
f=$(find $O/prof_decomp -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/decomp_kernel_stats.csv && head -14 $O/decomp_kernel_stats.csv | cut -c1-220
