#!/bin/bash
# round 3, job i: explicit FMAs in the marching kernels -- is the last bit now independent of chunk parity / instantiation? + perf check
cd /root/repo; mkdir -p gpurun_out/r3i
( time timeout 300 python tools/diag_bitexact2.py ) 2>&1 | grep -v "^Solution" | cut -c1-400 > gpurun_out/r3i/diag2.txt
timeout 200 python tools/tail_probe.py quick > gpurun_out/r3i/tail_quick.txt 2>&1
for w in "iso3dfd" "iso3dfd --size 512" "3axis" "3axis --size 1024"; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline --traffic none > "gpurun_out/r3i/bench_$(echo $w | tr ' -' '__').json" 2> /dev/null
done
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3i/pytest.log 2>&1
tail -3 gpurun_out/r3i/pytest.log
cat gpurun_out/r3i/diag2.txt gpurun_out/r3i/tail_quick.txt | cut -c1-300
for f in gpurun_out/r3i/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["frac"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
