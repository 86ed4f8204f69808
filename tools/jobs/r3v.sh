#!/bin/bash
# GPU job r3v: kernel trace of all six schedules of the 512^3 case in one process (the later ones were slow in r3t: what is running?)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3v; mkdir -p $O; cd $R
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -f csv -d $O/prof -- python $R/tools/overlap_probe.py --stencil iso3dfd --steps 20 --cases 1 ) > $O/prof.log 2>&1; echo "rocprof rc=$?"
grep '^{' $O/prof.log | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['schedule'][:34].ljust(34), r['ms_per_step'], 'ext', r['exterior_ms'], 'int', r['interior_ms'], 'pack', r['pack_ms'], 'copy', r['copy_ms'], 'unpack', r['unpack_ms'], 'wait', r['exposed_wait_ms'])
"
