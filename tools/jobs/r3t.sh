#!/bin/bash
# GPU job r3t: mirror transport with a co-residable copy kernel and an emulated link: all schedules at no link / 100 / 50 GB/s; tests
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3t; mkdir -p $O; cd $R
for g in 0 100 50; do
  echo "== link $g GB/s"
  YASK_MIRROR_LINK_GBPS=$g timeout 300 python tools/overlap_probe.py --tag _link$g 2>&1 | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['case'][:24].ljust(24), r['schedule'][:34].ljust(34), r['ms_per_step'], r['one_rank_block_ms_per_step'], r['vs_one_rank_block'], 'ext', r['exterior_ms'], 'int', r['interior_ms'], 'pack', r['pack_ms'], 'copy', r['copy_ms'], 'unpack', r['unpack_ms'], 'wait', r['exposed_wait_ms'])
"
  cp gpurun_out/overlap_probe_iso3dfd_link$g.json $O/
done
timeout 900 python -m pytest tests/test_transport_gpu.py tests/test_decomposed_blocks_gpu.py tests/test_multirank_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; grep -n "passed\|failed" $O/pytest.log | tail -3
