#!/bin/bash
# GPU job r6zl: the reference's 9-read test stencils (test_3d 0.39, test_stages_3d 0.53 of 8 TB/s on the vector point kernel): would the
# plane-ring kernel serve their 8 far corner reads (experimental build: plane-ring shapes for parts with > 3 mixed reads)?  + counters.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6zl; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R YASK_HIP_LIB_DIR=$R/yask_amd/lib_x
for s in test_3d test_stages_3d test_boundary_3d; do
  timeout 300 python3 tools/sweep_variants.py --stencil $s --size 512 --part 0 --chunks 0 --reps 5 --check --out $O/sweep_$s.json > $O/sweep_$s.log 2>&1
  echo "== $s"; grep "^{'variant'" $O/sweep_$s.log | sed "s/'xchunk': 0, //; s/, 'gpoints.*//" | sort -t: -k3 -n | head -n 8; grep mismatches $O/sweep_$s.log | grep -v ": 0$" | head -3
done
timeout 600 python3 tools/variant_pmc.py --stencil test_3d --variant vecpt_v4_z256_y4_x1 --part 0 --out $O/pmc_test_3d > $O/pmc_test_3d.log 2>&1; tail -n 3 $O/pmc_test_3d.log
