#!/bin/bash
# GPU job r3y: the pipelined half-exchange schedule (-hip_halves) -- compute side (decomp_cost), full schedule under the mirror
# transport with no link / 50 GB/s (overlap_probe), then the whole GPU suite (which holds its bit-exactness tests)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3y; mkdir -p $O; cd $R
fmt='
import sys,json
for l in sys.stdin:
    r=json.loads(l)
    if "schedule" in r: print(r["case"][:22].ljust(22), r["schedule"][:30].ljust(30), r["ms_per_step"], r["one_rank_block_ms_per_step"], r["vs_one_rank_block"], "ext", r["exterior_ms"], "int", r["interior_ms"], "pack", r["pack_ms"], "copy", r["copy_ms"], "unpack", r["unpack_ms"], "wait", r["exposed_wait_ms"])
    else: print(r["case"][:30].ljust(30), r["config"][:50].ljust(50), r["shell_or_exterior_ms"], r["rest_or_interior_ms"], r["undivided_ms"], r["overhead"])
'
timeout 90 python tools/decomp_cost.py --stencil iso3dfd --cases 3 --reps 6 --configs "halves,pct55 (default),regular-launch" 2>&1 | tee $O/decomp_cost.log | grep '^{' | python -c "$fmt"
cp gpurun_out/decomp_cost_iso3dfd.json $O/ 2>/dev/null
for g in 50 0; do
  echo "== link $g GB/s"
  YASK_MIRROR_LINK_GBPS=$g timeout 90 python tools/overlap_probe.py --tag _link$g --schedules "halves,planned (rounds,inline" --steps 30 2>&1 | tee $O/overlap_link$g.log | grep '^{' | python -c "$fmt"
  cp gpurun_out/overlap_probe_iso3dfd_link$g.json $O/ 2>/dev/null
done
timeout 260 python -m pytest tests -m gpu -q --timeout 150 -p no:cacheprovider > $O/pytest_all.log 2>&1; tail -15 $O/pytest_all.log
