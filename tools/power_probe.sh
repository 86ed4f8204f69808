#!/bin/bash
# GPU box: is the headline kernel's "6.25 TB/s over the fabric" a bandwidth ceiling or a power ceiling?  
# Runs the default bench at several power caps (rocm-smi --setpoweroverdrive, if the box allows it) and prints ms/step with the
# sclk / power the sampler saw under load.  Restores the default cap.
O=${1:-gpurun_out/power}
mkdir -p $O
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | head -30 > $O/smi_before.txt
for cap in default 1200 1000 800 600; do
  if [ "$cap" != "default" ]; then
    rocm-smi --setpoweroverdrive $cap > $O/set_$cap.txt 2>&1 || { echo "cap $cap: rocm-smi refused (rc=$?)"; tail -2 $O/set_$cap.txt; continue; }
  fi
  timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-probe --opts "-hip_placement_trials 1" > $O/bench_$cap.log 2>&1
  grep '^{' $O/bench_$cap.log | python -c "
import sys, json
j = json.loads(sys.stdin.read()); u = j['device_state']['under_load']
print('cap', '$cap', 'ms/step', j['ms_per_step'], 'Gpts/s', j['value'], 'frac', j['roofline']['frac'], 'under load:', {k: u.get(k) for k in ('sclk_mhz', 'mclk_mhz', 'power_w', 'n_samples') if k in u} or u)
"
done
rocm-smi --resetpoweroverdrive > $O/reset.txt 2>&1 || true
rocm-smi --showpower --showmaxpower 2>&1 | head -12 > $O/smi_after.txt
