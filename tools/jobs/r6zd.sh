#!/bin/bash
# GPU job r6zd: late refill (_lo) with trips / halo rings / 8-byte lanes / stores inside eval (_is) on awp, awp_elastic, ssg2.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6zd; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R YASK_HIP_LIB_DIR=$R/yask_amd/lib_x
for s in awp awp_elastic ssg2; do
  for p in 0 1; do
    timeout 400 python3 tools/sweep_variants.py --stencil $s --size 512 --part $p --chunks 0 --reps 5 --out $O/sweep_${s}_p$p.json > $O/sweep_${s}_p$p.log 2>&1
    echo "== $s part $p"; grep "^{'variant'" $O/sweep_${s}_p$p.log | sed "s/'xchunk': 0, //; s/, 'gpoints.*//" | sort -t: -k3 -n | head -n 14
  done
done
