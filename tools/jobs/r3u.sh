#!/bin/bash
# GPU job r3u: time line of the serial and slab schedules on the new mirror transport (why are they slow?)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3u; mkdir -p $O; cd $R
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -f csv -d $O/prof -- python $R/tools/overlap_probe.py --stencil iso3dfd --steps 20 --cases 1 --schedules "whole box" ) > $O/prof.log 2>&1; echo "rocprof rc=$?"
tail -3 $O/prof.log | cut -c1-300
