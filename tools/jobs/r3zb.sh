#!/bin/bash
# GPU job r3zb: rocprofv3 kernel trace of a rank on the halves schedule (mirror transport, 50 GB/s link): the two half-launches of the
# 242-VGPR twin, pack / copy / hold / unpack kernels
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3zb; mkdir -p $O; cd $R
( cd /tmp && export TMPDIR=/tmp && YASK_MIRROR_LINK_GBPS=50 timeout 45 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -- python $R/tools/overlap_probe.py --cases 1 --schedules halves --steps 30 --tag _prof ) > $O/prof.log 2>&1; echo "rc=$?"
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/halves_kernel_stats.csv && head -12 $O/halves_kernel_stats.csv | cut -c1-260
rm -rf $O/prof
grep '^{' $O/prof.log | cut -c1-400
