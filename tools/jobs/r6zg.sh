#!/bin/bash
# GPU job r6zg: fsg's stress part (23.8 of fsg's 30.3 ms) as four / eight clusters on the MARCHING kernel with one set of slabs (_sb).
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6zg; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R YASK_HIP_LIB_DIR=$R/yask_amd/lib_x
for p in 1 0; do
  timeout 900 python3 tools/sweep_variants.py --stencil fsg --size 512 --part $p --chunks 0 --reps 3 --check --out $O/sweep_fsg_p$p.json > $O/sweep_fsg_p$p.log 2>&1
  echo "== fsg part $p"; grep "^{'variant'" $O/sweep_fsg_p$p.log | sed "s/'xchunk': 0, //; s/, 'gpoints.*//" | sort -t: -k3 -n | head -n 14
  grep "mismatches" $O/sweep_fsg_p$p.log | sort | uniq -c | sort -rn | head -20
done
